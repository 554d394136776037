// 3x3 stride-1 "same" convolution of a 48- or 64-channel input into 64 channels (gfx950), weights in REGISTERS: the variant of conv3x3_res.hip that
// round 3's measurements ask for (profiles/r03z9_res3x3_timeline.txt, r03z11_res3x3_pmc.txt).  Tile 133, written at the very end of round 3: first GPU run
// bit-identical to tile 132 and 9-10 % faster (64 -> 64 at 320^2, bs 8: 87 -> 79 us, with a shortcut 114 -> 102 us; profiles/r03z14_rw3x3.txt); not tuned further.
//
// What the timeline of conv3x3_res shows: its 8 waves move through a tile's phases together -- a barrier (1 300 cycles of skew), a burst of DMA instructions at which
// every wave stalls (1 400), the MFMA loop (5 900, of which the matrix pipe needs 4 600) -- so each pipe idles while the others work, and the 72 KiB of weights in LDS
// leave no room for a second, independent block per CU whose phases would fill the gaps.  Here
//   * a block is 4 waves on an 8 x 16 tile; a wave owns 64 pixels (four tile rows) x 32 couts and keeps the 9 * CIN/16 weight fragments of ITS cout group in
//     registers for the whole kernel (144 VGPRs at CIN = 64: two waves per SIMD leave 256 each), so the LDS holds nothing but two 23 KiB patches per block;
//   * TWO such blocks are resident per CU (and a third fits the LDS): they synchronise only within themselves, drift apart, and one block's barrier, DMA issue and
//     SiLU epilogue overlap the other's MFMAs;
//   * LDS traffic per MFMA falls from 1.5 KiB (one activation + two weight fragments per two MFMAs) to 1 KiB (activation fragments only).
// Same patch layout (128-byte slots, pitch 18, chunk swizzle (u >> 1) & 7: conflict-free), K order and epilogue as conv3x3_res.hip: bit-identical results.
// Replaces yolort/v5/models/common.py:69-70,115-116 for Bottleneck(c, c).cv2 with c_ = 48 / 64.
#include "conv_common.hpp"
#include <cstdlib>

namespace ymi {

constexpr int RW_TH = 8, RW_TW = 16;                       // output tile
constexpr int RW_PH = RW_TH + 2, RW_PITCH = RW_TW + 2;     // patch rows / row pitch in slots
constexpr int RW_SLOTS = RW_PH * RW_PITCH;                 // 180 pixel slots of 128 B
constexpr int RW_PIECES = (RW_SLOTS * 8 + 63) / 64;        // 23 DMA pieces of 1 KiB
constexpr int RW_PPW = (RW_PIECES + 3) / 4;                // 6 pieces per wave
constexpr int RW_PATCH_BYTES = RW_PIECES * 1024;
constexpr int RW_J1 = 2 * RW_PITCH * 128;                  // byte distance of a wave's second pixel group (two tile rows down: the swizzle term repeats every 16 pixels)

template <int DT, int CIN>
__global__ __launch_bounds__(256, 2) void conv3x3_rw_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles) {
    typedef typename Mfma<DT>::frag frag;
    constexpr int KC = CIN / 16;            // k16 steps per tap
    constexpr int NCH = CIN / 8;            // real 16-byte chunks of a pixel (6 or 8)
    constexpr int NU = 9 * KC;              // (tap, k16) units: one weight fragment, two activation fragments, two MFMAs each
    static_assert(CIN == 48 || CIN == 64, "128-byte slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char rw_sm[];
    f32x4* bl = reinterpret_cast<f32x4*>(rw_sm);                    // bias [2 cout groups][4 octets][2 halves]
    unsigned char* patch0 = rw_sm + 256;                            // two patch buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;
    const int pg = wave >> 1, ct = wave & 1;      // pixel half of the tile (rows 4 pg .. 4 pg + 3), cout group

    // ---- this wave's weights: fragment (tap, kc) = rows ct*32 + frow, k = tap*CIN + kc*16 + hi*8 .. +7 (packed rows are zero padded to 128) ----
    frag wf[NU];
    {
        const uint16_t* wr = a.w + (int64_t)(ct * 32 + frow) * a.k_pad + hi * 8;
#pragma unroll
        for (int u = 0; u < NU; ++u) wf[u] = *reinterpret_cast<const frag*>(wr + u * 16);
    }
    if (tid < 16) {   // bias quad of (group t, octet g, half h): couts t*32 + g*8 + h*4 ..
        const int t = tid >> 3, g = (tid >> 1) & 3, h = tid & 1;
        bl[tid] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + g * 8 + h * 4);
    }

    // ---- patch DMA geometry (fixed per lane): entry e = piece*64 + lane -> slot e >> 3 = (pr, pc), position e & 7 holds chunk pos ^ v(pr, pc) ----
    int p_rc[RW_PPW];     // pr << 16 | chunk << 8 | pc, or -1: nothing to fetch
    int p_off[RW_PPW];
#pragma unroll
    for (int j = 0; j < RW_PPW; ++j) {
        int pi = wave * RW_PPW + j;
        pi = pi < RW_PIECES ? pi : RW_PIECES - 1;          // surplus slots re-send the last piece (identical bytes)
        const int e = pi * 64 + lane;
        const int q = e >> 3;
        const int qc = q < RW_SLOTS ? q : RW_SLOTS - 1;
        const int pr = qc / RW_PITCH, pc = qc - pr * RW_PITCH;
        const int chunk = (e & 7) ^ (((pr * RW_TW + pc) >> 1) & 7);
        p_rc[j] = (q < RW_SLOTS && chunk < NCH) ? ((pr << 16) | (chunk << 8) | pc) : -1;
        p_off[j] = (pr * a.w_in + pc) * a.x_cs + chunk * 8;
    }
    // ---- fragment geometry (fixed per lane): pixel group 0 of the wave: p = pg*64 + frow -> (r, c); group 1 is two tile rows below (+ RW_J1 bytes) ----
    const int pr_o = (pg * 64 + frow) / RW_TW, pc_o = (pg * 64 + frow) % RW_TW;
    int ea[9];            // byte offset of chunk `hi` (k16 step 0) of the tap's pixel; step kc: ^ (kc << 5)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int pr = pr_o + t / 3, pc = pc_o + t % 3;
        const int v = ((pr * RW_TW + pc) >> 1) & 7;
        ea[t] = (pr * RW_PITCH + pc) * 128 + ((hi ^ v) * 16);
    }

    auto tile_origin = [&](int idx, int& img, int& oy0, int& ox0) {
        int t = xcd_remap(idx, ntiles);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        img = t / tiles_y;
        oy0 = ty * RW_TH;
        ox0 = tx * RW_TW;
    };
    auto issue_patch = [&](int idx, unsigned char* dst) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        const bool interior = oy0 >= 1 && ox0 >= 1 && oy0 + RW_TH + 1 <= a.h && ox0 + RW_TW + 1 <= a.w_in;   // wave-uniform: the patch lies inside the image
        const int base = ((img * a.h + oy0 - 1) * a.w_in + ox0 - 1) * a.x_cs;
#pragma unroll
        for (int j = 0; j < RW_PPW; ++j) {
            int pi = wave * RW_PPW + j;
            pi = pi < RW_PIECES ? pi : RW_PIECES - 1;
            int off;
            if (interior) {
                off = p_rc[j] >= 0 ? base + p_off[j] : a.x_zero_off;
            } else {
                const int iy = oy0 - 1 + (p_rc[j] >> 16), ix = ox0 - 1 + (p_rc[j] & 0xff);
                const bool ok = p_rc[j] >= 0 && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
                off = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + ((p_rc[j] >> 8) & 0xff) * 8 : a.x_zero_off;
            }
            glds16(a.x + off, reinterpret_cast<uint16_t*>(dst + pi * 1024));
        }
    };

    int idx = blockIdx.x;
    int buf = 0;
    if (idx < ntiles) issue_patch(idx, patch0);
    for (; idx < ntiles; idx += gridDim.x) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // patch i has landed; everyone is done reading patch i-1 (first pass: the bias is written)
        const unsigned char* pb = patch0 + buf * RW_PATCH_BYTES;
        if (idx + (int)gridDim.x < ntiles) issue_patch(idx + gridDim.x, patch0 + (buf ^ 1) * RW_PATCH_BYTES);
        buf ^= 1;

        f32x16 acc[1][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = bl[(ct * 4 + g) * 2 + hi];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][0][g * 4 + e] = acc[0][1][g * 4 + e] = b[e];
        }
        // unit = (tap, k16): its two activation fragments are fetched under the previous unit's MFMAs; the tap's base address is laundered inside the loop (left alone,
        // the compiler hoists all 9 * KC addresses out of the tile loop: 36 registers this kernel does not have)
        frag fa[2][2];
        auto read_unit = [&](auto ut, auto bt) {
            constexpr int u = decltype(ut)::value, b = decltype(bt)::value;
            constexpr int t = u / KC, kc = u % KC;
            int eb = ea[t];
            asm volatile("" : "+v"(eb));
            const unsigned char* p0 = pb + (eb ^ (kc << 5));
            fa[b][0] = *reinterpret_cast<const frag*>(p0);
            fa[b][1] = *reinterpret_cast<const frag*>(p0 + RW_J1);
        };
        read_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, NU>([&](auto ut) {
            constexpr int u = decltype(ut)::value;
            if constexpr (u + 1 < NU) read_unit(std::integral_constant<int, u + 1>{}, std::integral_constant<int, (u + 1) & 1>{});
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[0][j] = Mfma<DT>::run(wf[u], fa[u & 1][j], acc[0][j]);
        });
        // the lean epilogue only (the launcher admits nothing else: SiLU, cout = 64, tensors below 2^31 elements): the general form's registers, on top of 144 for the
        // weights, spilled
        auto pix = [&](int j, int64_t& m, bool& ok) {
            const int oy = oy0 + pr_o + 2 * j, ox = ox0 + pc_o;
            ok = oy < a.ho && ox < a.wo;
            m = ((int64_t)img * a.ho + oy) * a.wo + ox;
        };
        if (a.res != nullptr) finish_wave_tile_lean<DT, 1, 2, true>(a, acc, ct * 32, hi, pix);
        else finish_wave_tile_lean<DT, 1, 2, false>(a, acc, ct * 32, hi, pix);
    }
}

template <int DT, int CIN>
static int launch_rw(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int tiles_x = cdiv(a.wo, RW_TW), tiles_y = cdiv(a.ho, RW_TH);
    const int ntiles = a.n * tiles_x * tiles_y;
    const size_t lds = 256 + (size_t)2 * RW_PATCH_BYTES;
    auto kfn = conv3x3_rw_kernel<DT, CIN>;
    int resident = 512;   // two 4-wave blocks per CU
    if (const char* e = getenv("YOLORT_AMD_RES3X3_BLOCKS")) {   // test aid: few blocks walk many tiles (the persistent loop on small inputs)
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) resident = v;
    }
    a.nblk_m = ntiles;
    a.nblk_n = 1;
    hipLaunchKernelGGL(kfn, dim3(ntiles < resident ? ntiles : resident), dim3(256), lds, s, a, tiles_x, tiles_y, ntiles);
    return check_launch("conv3x3_rw_kernel");
}

// variant 1 (the only one): cin selects the instantiation
int conv3x3_rw_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(variant == 1, "ymi_conv2d: unknown register-weights 3x3 variant %d", variant);
    YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.ph == 1 && a.pw == 1 && a.sh == 1 && a.sw == 1 && (a.cin == 48 || a.cin == 64) && a.k_pad >= 9 * a.cin &&
                    a.cout == 64 && a.cout_pad >= a.cout && a.zeros != nullptr && a.up2 == 0 && a.split == 0 && a.chain_w == nullptr && out_dtype == dtype && a.act == YMI_ACT_SILU,
                "ymi_conv2d: the register-weights 3x3 kernel (tile 133) handles cin = 48 / 64, cout = 64, stride 1, pad 1, SiLU, 16-bit output, no chained conv (and needs desc.zeros)");
    {
        const int64_t cs_max = a.y_cs > a.res_cs ? a.y_cs : a.res_cs;
        YMI_REQUIRE(((int64_t)a.M + 1) * cs_max < ((int64_t)1 << 31), "ymi_conv2d: tile 133: output / shortcut tensor too large for 32-bit offsets");
    }
    YMI_REQUIRE((int64_t)a.n * a.h * a.w_in * a.x_cs < ((int64_t)1 << 31), "ymi_conv2d: input tensor too large for 32-bit offsets");
    const bool f16 = dtype == YMI_F16;
    if (a.cin == 64) return f16 ? launch_rw<YMI_F16, 64>(a, s) : launch_rw<YMI_BF16, 64>(a, s);
    return f16 ? launch_rw<YMI_F16, 48>(a, s) : launch_rw<YMI_BF16, 48>(a, s);
}

}  // namespace ymi
