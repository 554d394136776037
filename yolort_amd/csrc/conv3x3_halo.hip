// 3x3 stride-1 "same" convolution with an LDS-resident input halo (gfx950).
//
// Why: the implicit-GEMM kernels (conv_igemm.hip) fetch every activation 9 times (once per tap)
// through L1 into LDS.  Measured on MI355X the global->LDS fill path saturates at roughly
// 13 B/clk/CU (~7 TB/s chip-wide) however the loads are issued, and the 3x3 layers run exactly at
// (im2col bytes + weight re-reads) / 7 TB/s.  Here a block owns an 8x16 output patch of one image:
// per 32-channel chunk the (8+2)x(16+2) input patch is brought into LDS ONCE (12 KiB, LDS-DMA,
// double buffered) and the nine taps read their MFMA fragments from it at shifted addresses, so the
// activation fill traffic drops from 9x to 1.4x; only the weight tiles still stream per (tap,chunk)
// step through the same DMA ring / counted-vmcnt / one-barrier-per-step pipeline as the v2 kernel.
//
// Same math as conv_igemm (swapped MFMA D[cout][pixel], fp32 accumulate, bias+SiLU(+residual)
// epilogue, channel-slice views).  Replaces yolort/v5/models/common.py:69-70,115-116 for the
// Bottleneck.cv2 convolutions (k=3, s=1, p=1, cin % 32 == 0).
#include "conv_common.hpp"

namespace ymi {

constexpr int HTH = 8, HTW = 16;             // output patch
constexpr int HPH = HTH + 2, HPW = HTW + 2;  // input patch (halo 1)
constexpr int HPIX = 192;                    // patch pixels per LDS buffer (180 used, 12 DMA pieces of 16)
constexpr int HBM = HTH * HTW;               // 128 output pixels per block

template <int DT, int ODT, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    static_assert((HBM / WM) * (BN / WN) == 4, "4 waves per block");
    static_assert(STAGES == 2 || STAGES == 3, "weight ring depth");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int W_PIECES = BN / 16;
    constexpr int PWV = (W_PIECES + 3) / 4;          // weight pieces per wave per step
    constexpr int PATCH_HALFS = HPIX * 32;
    constexpr int WSTAGE_HALFS = BN * 32;
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [2 patch buffers][W ring]
    uint16_t* wring = smem + 2 * PATCH_HALFS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = (wave / WAVES_N) * WM, wave_n = (wave % WAVES_N) * WN;

    const int nblk = a.nblk_m * a.nblk_n;
    const int lb = xcd_remap(blockIdx.x, nblk);
    const int bn = lb % a.nblk_n;
    int t = lb / a.nblk_n;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int img = t / tiles_y;
    const int oy0 = ty * HTH, ox0 = tx * HTW, n0 = bn * BN;
    const int nchunks = a.cin / 32;
    const int nsteps = nchunks * 9;

    // ---- patch DMA geometry: 12 pieces of 16 pixels, 3 per wave; lane (pixel q, position pos) fetches
    //      k-chunk pos ^ ((q>>2)&3) of input pixel (oy0-1+q/18, ox0-1+q%18), or the zero page ----
    int p_off[3];
    int p_slot[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pi = wave * 3 + j;
        const int q = pi * 16 + (lane >> 2);
        p_slot[j] = pi * 512;
        const int qc = q < HPH * HPW ? q : HPH * HPW - 1;
        const int pr = qc / HPW, pc = qc - pr * HPW;
        const int iy = oy0 - 1 + pr, ix = ox0 - 1 + pc;
        const bool ok = (q < HPH * HPW) && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
        const int kchunk = (lane & 3) ^ ((q >> 2) & 3);
        p_off[j] = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + kchunk * 8 : -1;
    }
    // ---- weight DMA geometry (as conv_igemm v2: rows are zero padded to 128) ----
    const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);
    int w_off[PWV], w_slot[PWV];
#pragma unroll
    for (int j = 0; j < PWV; ++j) {
        int pi = wave * PWV + j;
        pi = pi < W_PIECES ? pi : W_PIECES - 1;
        w_slot[j] = pi * 512;
        w_off[j] = (n0 + pi * 16 + (lane >> 2)) * a.k_pad + wchunk * 8;
    }

    auto issue_patch = [&](int chunk) {
        uint16_t* dst = smem + (chunk & 1) * PATCH_HALFS;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int off = p_off[j] >= 0 ? p_off[j] + chunk * 32 : a.x_zero_off;
            glds16(a.x + off, dst + p_slot[j]);
        }
    };
    auto issue_w = [&](int step, int chunk, int tap) {   // weights of k = tap*cin + chunk*32 .. +31
        uint16_t* dst = wring + (step % STAGES) * WSTAGE_HALFS;
        const int koff = tap * a.cin + chunk * 32;
#pragma unroll
        for (int j = 0; j < PWV; ++j) glds16(a.w + (w_off[j] + koff), dst + w_slot[j]);
    };

    f32x16 acc[TN][TM];

    // issue-side position (chunk, tap) of the next weight step to fetch
    int is_step = 0, is_chunk = 0, is_tap = 0;
    auto issue_next_w = [&]() {
        issue_w(is_step, is_chunk, is_tap);
        ++is_step;
        if (++is_tap == 9) { is_tap = 0; ++is_chunk; }
    };

    issue_patch(0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) issue_next_w();
    f32x4 bias_regs[TN][4];   // issued behind the prologue DMA (conv_common.hpp)
    load_bias<TN>(a, n0 + wave_n, lane >> 5, bias_regs);
    init_acc<TN, TM>(acc, bias_regs);   // accumulate on top of the bias
    // (after the prologue DMA issue: waiting for the bias load first put two cold memory latencies in series at every block start)

    // per-lane fragment geometry
    const int frow = lane & 31;
    const int hi = lane >> 5;
    int q0[TM];   // patch pixel of this lane's output pixel for tap (0,0): (row + 0)*18 + col + 0
#pragma unroll
    for (int j = 0; j < TM; ++j) q0[j] = (2 * (wave_m / 32 + j) + ((lane >> 4) & 1)) * HPW + (lane & 15);
    const int wswz = (lane >> 2) & 3;
    const int wpos0 = ((0 + hi) ^ wswz) * 8, wpos1 = ((2 + hi) ^ wswz) * 8;

    int chunk = 0, tap = 0, dy = 0, dx = 0;
    bool patch_recent = false;   // a patch (3 pieces) was issued during the previous step
    for (int step = 0; step < nsteps; ++step) {
        // weights of `step` (and, in queue order before them, every patch issued earlier) have landed
        // once at most the loads issued after them are still pending
        // Loads still allowed in flight = those issued AFTER the awaited weights: with a 3-deep ring that
        // is what the previous step issued (its optional patch, 3 pieces, then one weight step).
        if constexpr (STAGES == 2) {
            wait_vmcnt<0>();
        } else {
            const bool next_w = step + 1 < nsteps;
            if (next_w && patch_recent) wait_vmcnt<PWV + 3>();
            else if (next_w) wait_vmcnt<PWV>();
            else if (patch_recent) wait_vmcnt<3>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        patch_recent = false;
        if (tap == 0 && chunk + 1 < nchunks) {   // everyone is done with patch buffer (chunk+1)&1 (chunk-1's)
            issue_patch(chunk + 1);
            patch_recent = true;
        }
        if (step + STAGES - 1 < nsteps) issue_next_w();

        const uint16_t* pb = smem + (chunk & 1) * PATCH_HALFS;
        const uint16_t* ws = wring + (step % STAGES) * WSTAGE_HALFS + wave_n * 32;
        const int tapoff = dy * HPW + dx;
        frag af[2][TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int q = q0[j] + tapoff;
            const int swz = (q >> 2) & 3;
            const int e0 = q * 32 + ((hi ^ swz) * 8);
            af[0][j] = *reinterpret_cast<const frag*>(pb + e0);
            af[1][j] = *reinterpret_cast<const frag*>(pb + (e0 ^ 16));   // k-chunk (2+hi)^swz = flip bit 1
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag wf[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const frag*>(ws + (i * 32 + frow) * 32 + (ks ? wpos1 : wpos0));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = Mfma<DT>::run(wf[i], af[ks][j], acc[i][j]);
        }
        // advance (chunk, tap)
        if (++tap == 9) { tap = 0; dy = 0; dx = 0; ++chunk; }
        else if (++dx == 3) { dx = 0; ++dy; }
    }

    // ---- epilogue: SiLU (+ residual), 16-byte stores straight from the MFMA layout (conv_common.hpp) ----
    auto pix = [&](int j, int64_t& m, bool& ok) {
        const int oy = oy0 + 2 * (wave_m / 32 + j) + ((lane >> 4) & 1), ox = ox0 + (lane & 15);
        ok = oy < a.ho && ox < a.wo;
        m = ((int64_t)img * a.ho + oy) * a.wo + ox;
    };
    if constexpr (ODT == DT && BN == WN && TN <= 2) {   // pixel-major waves: a chained 1x1 (C3.cv3) can run from the outputs in registers
        if (a.chain_w != nullptr) {
            finish_wave_tile_chain<DT, TN, TM>(a, acc, lane >> 5, lane, pix);
            return;
        }
    }
    finish_wave_tile<DT, ODT, TN, TM>(a, acc, n0 + wave_n, lane >> 5, pix);
}

template <int DT, int ODT, int BN, int WM, int WN, int STAGES>
static int launch_halo(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int tiles_x = cdiv(a.wo, HTW), tiles_y = cdiv(a.ho, HTH);
    a.nblk_m = a.n * tiles_x * tiles_y;
    a.nblk_n = cdiv(a.cout_pad, BN);
    size_t lds = (size_t)2 * HPIX * 64 + (size_t)STAGES * BN * 64;
    auto kfn = conv3x3_halo_kernel<DT, ODT, BN, WM, WN, STAGES>;
    if (lds < lds_floor_bytes()) lds = lds_floor_bytes();
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    hipLaunchKernelGGL(kfn, dim3(a.nblk_m * a.nblk_n), dim3(256), lds, s, a, tiles_x, tiles_y);
    return check_launch("conv3x3_halo_kernel");
}

template <int DT, int ODT>
static int halo_variant(const ConvArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        case 1: return launch_halo<DT, ODT, 128, 64, 64, 3>(a, s);   // 2x2 waves of 64 px x 64 cout
        case 2: return launch_halo<DT, ODT, 64, 64, 32, 3>(a, s);    // 2x2 waves of 64 px x 32 cout
        case 3: return launch_halo<DT, ODT, 32, 32, 32, 3>(a, s);    // 4x1 waves of 32 px x 32 cout
        case 4: return launch_halo<DT, ODT, 128, 64, 64, 2>(a, s);
        case 5: return launch_halo<DT, ODT, 64, 64, 32, 2>(a, s);
        case 6: return launch_halo<DT, ODT, 32, 32, 32, 2>(a, s);
        case 7: return launch_halo<DT, ODT, 64, 32, 64, 3>(a, s);    // 4x1 waves of 32 px x 64 cout
        default: set_error("ymi_conv2d: unknown halo variant %d", variant); return YMI_EINVAL;
    }
}

int conv3x3_halo_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.sh == 1 && a.sw == 1 && a.ph == 1 && a.pw == 1 && a.cin % 32 == 0 && a.zeros != nullptr && a.split == 0,
                "ymi_conv2d: the LDS-halo kernel handles 3x3 stride-1 pad-1 convolutions with cin %% 32 == 0 (and needs desc.zeros)");
    YMI_REQUIRE(a.k_pad == 9 * a.cin, "ymi_conv2d: halo kernel expects k_pad == 9*cin");
    if (a.chain_w != nullptr) {   // chained conv: pixel-major variants only, cout width == the chain's fresh K
        const int bn = (variant == 3 || variant == 6) ? 32 : (variant == 7 ? 64 : 0);
        YMI_REQUIRE(bn != 0 && bn == a.chain_k && a.cout_pad == bn, "ymi_conv2d: this halo variant does not fit the chained convolution (cout width must equal %d)", a.chain_k);
    }
    if (dtype == YMI_F16) return out_dtype == YMI_F32 ? halo_variant<YMI_F16, YMI_F32>(a, variant, s) : halo_variant<YMI_F16, YMI_F16>(a, variant, s);
    return out_dtype == YMI_F32 ? halo_variant<YMI_BF16, YMI_F32>(a, variant, s) : halo_variant<YMI_BF16, YMI_BF16>(a, variant, s);
}

}  // namespace ymi
