// 3x3 stride-1 "same" convolution, 8-wave LDS-halo kernel (gfx950) -- second generation of conv3x3_halo.hip.
//
// What bounded the 4-wave kernels on these layers: per-wave instruction issue.  A
// `global_load_lds_dwordx4` costs its wave ~60-180 cycles to issue against 32 for an MFMA, and a 128x128x32 step of the
// implicit-GEMM kernel is 4 DMA pieces per wave against 8 MFMAs.  Here
//   * a block is 8 waves (2 per SIMD) on a TH x TW output patch of up to 256 pixels of ONE image; TH, TW are chosen per
//     feature-map size on the host so that the patches tile the map with little waste (16x16 on 80x80, 10x20 on 20x20 and
//     40x40, ...);
//   * per 32-channel chunk the (TH+2) x (TW+2) input patch goes to LDS ONCE (<= 24 DMA pieces, double buffered) and feeds all
//     nine taps; only the weights stream, and they stream in LONG steps: one step = one kernel ROW (3 taps) of one chunk =
//     3 x BN x 64 B.  A wave issues <= 3 weight pieces + 1 patch piece per step against 24 MFMAs (BN = 128), spread between
//     the MFMA groups; there is ONE barrier per 24 MFMAs per wave (the first version of this kernel synchronised per tap:
//     8 MFMAs per barrier, and both waves of a SIMD sat in DMA issue / ds_read latency at the same time -- measured 880
//     cycles of "compute" per 256 cycles of MFMA);
//   * with steps of >= 1.5k cycles a two-deep weight ring and a plain vmcnt(0) suffice (loads have a whole step to land);
//   * inside a step the fragments of sub-step i+1 (tap, k16 half) are fetched under the MFMAs of sub-step i.
// One raw s_barrier per step / source-side XOR swizzle exactly as in conv_igemm_impl.hpp (v2).
//
// Same arithmetic and accumulator layout as every other conv kernel of this library (swapped MFMA D[cout][pixel], fp32
// accumulate on top of the folded-BN bias, SiLU (+ residual) epilogue of conv_common.hpp, channel-slice views).
// Replaces yolort/v5/models/common.py:69-70,115-116 for the Bottleneck.cv2 convolutions (k=3, s=1, p=1, cin % 32 == 0).
#include "conv_common.hpp"
#include <cstdio>
#include <cstdlib>

namespace ymi {

#ifdef YMI_STAMPS   // tuning aid (never in the shipped build): s_memtime timeline, wave 0 of each block (tools/stamp_conv.py)
__device__ unsigned long long ymi_stamps_h8[2048 * 128];
#define H8_STAMP(i)                                                                                                       \
    do {                                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 2048 && (i) < 128) ymi_stamps_h8[blockIdx.x * 128 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define H8_STAMP(i) ((void)0)
#endif

struct Halo8Geom {
    int th, tw;              // output patch (th * tw <= 256)
    int pw;                  // patch row pitch in LDS: tw + 4 (tw % 4 == 0; columns tw + 2, tw + 3 are padding, never read)
    int twq;                 // tw / 4
    int ppix;                // (th + 2) * pw patch slots
    int ppieces;             // ceil(ppix / 16) DMA pieces per patch chunk (<= 24)
    int tiles_x, tiles_y;
    unsigned magic_tw, magic_pw;
};

template <int DT, int ODT, int BN, int WAVES_M>
__global__ __launch_bounds__(512, 2) void conv_halo8_kernel(const ConvArgs a, const Halo8Geom g) {
    constexpr int WAVES_N = 8 / WAVES_M;
    constexpr int WM = 256 / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WAVES_M * WAVES_N == 8, "8 waves");
    constexpr int ROW_PIECES = BN / 16;              // weight pieces of one tap
    constexpr int W_PIECES = 3 * ROW_PIECES;         // ... of one stage = one kernel row (3 taps) x 32 channels
    constexpr int PW = (W_PIECES + 7) / 8;           // weight pieces per wave per step
    constexpr int WSTAGE_HALFS = 3 * BN * 32;
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [patch buffer 0][patch buffer 1 (cin > 32)][weight stages 0, 1, 2]
    const int patch_halfs = g.ppieces * 512;
    uint16_t* wring = smem + (a.cin > 32 ? 2 : 1) * patch_halfs;   // a single 32-channel chunk needs no second patch buffer (more blocks per CU)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = (wave / WAVES_N) * WM, wave_n = (wave % WAVES_N) * WN;

    const int nblk = a.nblk_m * a.nblk_n;
    const int lb = xcd_remap(blockIdx.x, nblk);
    const int bn = lb % a.nblk_n;
    int t = lb / a.nblk_n;
    const int tx = t % g.tiles_x;
    t /= g.tiles_x;
    const int ty = t % g.tiles_y;
    const int img = t / g.tiles_y;
    const int oy0 = ty * g.th, ox0 = tx * g.tw, n0 = bn * BN;
    const int nchunks = a.cin / 32;
    H8_STAMP(0);
#ifdef YMI_STAMPS   // slot 126: where the block runs (HW_ID: wave / simd / cu / sh / se; XCC_ID)
    if (threadIdx.x == 0 && blockIdx.x < 2048) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ymi_stamps_h8[blockIdx.x * 128 + 126] = ((unsigned long long)xcc << 32) | hw;
        ymi_stamps_h8[blockIdx.x * 128 + 124] = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz counter: ticks of slot 0 / 127 vs these give the shader clock
    }
#endif

    // ---- patch DMA geometry: piece pi = 16 patch slots; wave w owns pieces w, w+8, w+16 (clamped: surplus slots re-send
    //      the last piece, identical bytes).  Lane (slot q = pr * pw + pc, position pos) fetches k-chunk pos ^ v(pr, pc) of input pixel
    //      (oy0-1+pr, ox0-1+pc), or the zero page outside the image / in the padding columns / past the patch.
    //      LAYOUT (round 3): a `ds_read_b128` is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
    //      (MI355X_MICROARCH.md, LDS) -- at one cycle per group when the 16 lanes hit 16 different 16-byte slots of the 256-byte bank row.  A wave's 32
    //      pixels are consecutive in TILE order u = r * tw + c, every tap shifts all of them by one constant, and each group's lanes cover all residues
    //      mod 16 -- so the slot of a fragment, (q & 3) * 4 + pos, must be a bijection of u mod 16.  With the dense pitch tw + 2 it was not (q = u + 2 r: every
    //      row crossing inside the 32 pixels shifts the residues by 2 and two lanes collide on two slots: 7-8 cycles per activation read instead
    //      of 4 on every patch shape but 8 x 32; measured as 38 % LDS bank-conflict cycles).  With pitch tw + 4 and tw % 4 == 0, q & 3 = u & 3, and the
    //      chunk swizzle v = (u >> 2) & 3 = (pr * tw/4 + pc/4) & 3 supplies the other two bits: 4 cycles for every shape and tap. ----
    int p_off[3], p_slot[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int pi = wave + 8 * j;
        pi = pi < g.ppieces ? pi : g.ppieces - 1;
        const int q = pi * 16 + (lane >> 2);
        p_slot[j] = pi * 512;
        const int qc = q < g.ppix ? q : g.ppix - 1;
        const int pr = fast_div(qc, g.pw, g.magic_pw), pc = qc - pr * g.pw;
        const int iy = oy0 - 1 + pr, ix = ox0 - 1 + pc;
        const bool ok = (q < g.ppix) && (pc < g.tw + 2) && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
        const int kchunk = (lane & 3) ^ ((pr * g.twq + (pc >> 2)) & 3);
        p_off[j] = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + kchunk * 8 : -1;
    }
    // ---- weight DMA geometry: stage = [tap 0..2][BN cout rows] x 64 B; wave w moves pieces w, w+8, ... (clamped);
    //      packed weight rows are zero padded to 128 ----
    const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);
    int w_off[PW], w_slot[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        int pi = wave + 8 * j;
        pi = pi < W_PIECES ? pi : W_PIECES - 1;
        const int tap = pi / ROW_PIECES, rp = pi - tap * ROW_PIECES;
        w_slot[j] = pi * 512;
        w_off[j] = (n0 + rp * 16 + (lane >> 2)) * a.k_pad + tap * a.cin + wchunk * 8;
    }

    auto issue_patch_piece = [&](int chunk, auto jt) {
        constexpr int j = decltype(jt)::value;
        uint16_t* dst = smem + (chunk & 1) * patch_halfs;
        const int off = p_off[j] >= 0 ? p_off[j] + chunk * 32 : a.x_zero_off;
        glds16(a.x + off, dst + p_slot[j]);
    };
    // weights of kernel row dy, channels chunk*32 .. +31 (k = (dy*3 + tap)*cin + chunk*32 + c) into stage `slot`
    auto issue_w_piece = [&](int kbase, int slot, auto jt) {
        constexpr int j = decltype(jt)::value;
        glds16(a.w + (w_off[j] + kbase), wring + slot * WSTAGE_HALFS + w_slot[j]);
    };

    f32x16 acc[TN][TM];

    // prologue: the whole first patch (3 pieces per wave) and weight stage 0
    static_for<0, 3>([&](auto jt) { issue_patch_piece(0, jt); });
    static_for<0, PW>([&](auto jt) { issue_w_piece(0, 0, jt); });
    static_for<0, PW>([&](auto jt) { issue_w_piece(3 * a.cin, 1, jt); });   // stage 1 = (chunk 0, dy 1): nsteps >= 3 always
    f32x4 bias_regs[TN][4];   // issued behind the prologue DMA (conv_common.hpp)
    load_bias<TN>(a, n0 + wave_n, lane >> 5, bias_regs);
    init_acc<TN, TM>(acc, bias_regs);   // accumulate on top of the bias
    // (after the prologue DMA issue: waiting for the bias load first put two cold memory latencies in series at every block start)

    // ---- per-lane fragment geometry: LDS byte offsets of the nine taps' activation fragments (k16 half 0; half 1 = ^ 32),
    //      relative to the current patch buffer -- computed once, not per sub-step (the q -> swizzle arithmetic was ~6 VALU
    //      per fragment read, 970 VALU per wave against 144 MFMAs on a 128 -> 128 layer) ----
    const int frow = lane & 31;
    const int hi = lane >> 5;
    const int npix = g.th * g.tw;
    int ea[TM][9];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int p = wave_m + j * 32 + frow;
        const int pc = p < npix ? p : 0;
        const int r = fast_div(pc, g.tw, g.magic_tw), c = pc - r * g.tw;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pr = r + t / 3, pcc = c + t % 3;
            const int q = pr * g.pw + pcc;
            ea[j][t] = (q * 32 + ((hi ^ ((pr * g.twq + (pcc >> 2)) & 3)) * 8)) * 2;
        }
    }
    const int wswz = (lane >> 2) & 3;
    // weight fragments: byte offset of (row frow, k16 half ks) inside a stage, relative to wring + wave_n rows
    const int wb0 = ((wave_n + frow) * 32 + ((0 + hi) ^ wswz) * 8) * 2;
    const int wb1 = ((wave_n + frow) * 32 + ((2 + hi) ^ wswz) * 8) * 2;
    const unsigned char* const lds = reinterpret_cast<const unsigned char*>(smem);
    const int wring_b = (int)((a.cin > 32 ? 2 : 1) * patch_halfs) * 2;
    const int patch_b = patch_halfs * 2;

    // One step = one kernel row dy (3 taps) of one 32-channel chunk; stage index s = 3*chunk + dy, so the weight ring slot of a
    // step is dy itself (three-deep ring).  The stage of step s+2 is issued during step s: with a two-deep ring the loads of
    // the next stage had ONE step (~2 k cycles) to come back from L2 and the top-of-step wait + barrier cost 720 of the
    // 2280 cycles of a step (measured, 128 -> 128 at 40x40, two blocks per CU).
    int pbuf = 0;   // byte offset of the current patch buffer
    auto do_step = [&](auto dyt, int chunk, bool more_chunks) {
        constexpr int dy = decltype(dyt)::value;
        // what may still be in flight when stage s (and, at dy == 0, this chunk's patch) must have landed: the pieces this wave
        // issued during the previous step after them -- PW weight pieces of stage s+1 (if there is one) and, inside a chunk,
        // the patch piece of the next chunk (issued BEFORE those weights at dy >= 1; at dy == 0 it is awaited itself)
        if constexpr (dy == 0) {
            wait_vmcnt<PW>();
        } else if constexpr (dy == 1) {
            if (more_chunks) wait_vmcnt<PW + 1>(); else wait_vmcnt<PW>();
        } else {
            if (more_chunks) wait_vmcnt<PW + 1>(); else wait_vmcnt<0>();
        }
        H8_STAMP(4 + (chunk * 3 + dy) * 3);
        __builtin_amdgcn_s_barrier();          // every wave's pieces of this stage landed; everyone is done with stage s-1
        __builtin_amdgcn_sched_barrier(0);
        H8_STAMP(5 + (chunk * 3 + dy) * 3);
        // stage s+2: (chunk, 2) | (chunk+1, 0) | (chunk+1, 1), ring slot (dy + 2) % 3
        constexpr int ndy = (dy + 2) % 3;
        const bool issue_w = dy == 0 ? true : more_chunks;
        const int nkbase = ndy * 3 * a.cin + (dy == 0 ? chunk : chunk + 1) * 32;
        const unsigned char* pb = lds + pbuf;
        const unsigned char* ws = lds + wring_b + dy * (WSTAGE_HALFS * 2);
        frag fa[2][TM], fw[2][TN];
        auto read_frags = [&](auto subt, auto buft) {   // sub-step = (tap, k16 half)
            constexpr int sub = decltype(subt)::value, buf = decltype(buft)::value;
            constexpr int tap = sub >> 1, ks = sub & 1;
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[buf][j] = *reinterpret_cast<const frag*>(pb + (ks ? (ea[j][dy * 3 + tap] ^ 32) : ea[j][dy * 3 + tap]));
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[buf][i] = *reinterpret_cast<const frag*>(ws + (ks ? wb1 : wb0) + (tap * BN + i * 32) * 64);
        };
        read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, 6>([&](auto subt) {
            constexpr int sub = decltype(subt)::value;
            // fragments of the next sub-step first (their latency hides under this sub-step's MFMAs) ...
            if constexpr (sub + 1 < 6) read_frags(std::integral_constant<int, sub + 1>{}, std::integral_constant<int, (sub + 1) & 1>{});
            // ... then one slice of the DMA issue (spread over the sub-steps instead of a burst at the step's head)
            if constexpr (sub == 0) {
                if (more_chunks) issue_patch_piece(chunk + 1, std::integral_constant<int, dy>{});
            } else if constexpr (sub - 1 < PW) {
                if (issue_w) issue_w_piece(nkbase, ndy, std::integral_constant<int, sub - 1>{});
            }
            YMI_PRIO_HI();
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = Mfma<DT>::run(fw[sub & 1][i], fa[sub & 1][j], acc[i][j]);
            YMI_PRIO_LO();
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave is done reading stage s before it reaches the next barrier
        H8_STAMP(6 + (chunk * 3 + dy) * 3);
    };
    H8_STAMP(1);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more_chunks = chunk + 1 < nchunks;
        do_step(std::integral_constant<int, 0>{}, chunk, more_chunks);
        do_step(std::integral_constant<int, 1>{}, chunk, more_chunks);
        do_step(std::integral_constant<int, 2>{}, chunk, more_chunks);
        pbuf = patch_b - pbuf;   // the other patch buffer (a single-chunk launch never gets here twice)
    }

    // ---- epilogue: SiLU (+ residual), 16-byte stores straight from the MFMA layout (conv_common.hpp) ----
    H8_STAMP(3);
    auto pix = [&](int j, int64_t& m, bool& ok) {
        const int p = wave_m + j * 32 + frow;
        const int pc = p < npix ? p : 0;
        const int r = fast_div(pc, g.tw, g.magic_tw), c = pc - r * g.tw;
        const int oy = oy0 + r, ox = ox0 + c;
        ok = p < npix && oy < a.ho && ox < a.wo;
        m = ((int64_t)img * a.ho + oy) * a.wo + ox;
    };
    if constexpr (ODT == DT && WAVES_M == 8 && TM == 1 && TN <= 4 && TN != 3) {   // 8 x 1 waves: a wave owns ALL couts of its 32 pixels -- a chained 1x1 (the next
        if (a.chain_w != nullptr) {                                     // Bottleneck's cv1, or C3.cv3) runs from the outputs in registers
            finish_wave_tile_chain<DT, TN, TM>(a, acc, lane >> 5, lane, pix);
            H8_STAMP(127);
            return;
        }
    }
    finish_wave_tile<DT, ODT, TN, TM>(a, acc, n0 + wave_n, lane >> 5, pix);
    H8_STAMP(127);
#ifdef YMI_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 2048) ymi_stamps_h8[blockIdx.x * 128 + 125] = __builtin_amdgcn_s_memrealtime();
#endif
}

// patch shape for a ho x wo map: th * tw <= 256, tw % 4 == 0 (the conflict-free LDS layout), (th+2) * (tw+4) <= 24 * 16 patch slots; maximise the
// fraction of the 256 lanes that carry real output pixels (tile overhang counts as waste), then prefer the smaller halo
static void choose_patch(int ho, int wo, int& th_best, int& tw_best) {
    if (const char* e = getenv("YOLORT_AMD_H8_PATCH")) {   // tuning aid: "th,tw"
        int th = 0, tw = 0;
        if (sscanf(e, "%d,%d", &th, &tw) == 2 && th >= 1 && tw >= 4 && tw % 4 == 0 && th * tw <= 256 && (th + 2) * (tw + 4) <= 24 * 16) { th_best = th; tw_best = tw; return; }
    }
    double best = -1.0;
    th_best = 16; tw_best = 16;
    for (int tw = 4; tw <= 64 && tw <= ((wo + 3) / 4) * 4; tw += 4) {
        int th = 256 / tw;
        if (th > ho) th = ho;
        for (int thc = th; thc >= 1 && thc >= th - 8; --thc) {
            if ((thc + 2) * (tw + 4) > 24 * 16) continue;
            const int ty = (ho + thc - 1) / thc, tx = (wo + tw - 1) / tw;
            const double util = (double)ho * wo / ((double)ty * tx * 256.0);
            const double halo = (double)(thc + 2) * (tw + 2) / ((double)thc * tw);
            const double score = util - 0.02 * halo;
            if (score > best) { best = score; th_best = thc; tw_best = tw; }
        }
    }
}

template <int DT, int ODT, int BN, int WAVES_M>
static int launch_halo8(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    Halo8Geom g;
    choose_patch(a.ho, a.wo, g.th, g.tw);
    g.pw = g.tw + 4;
    g.twq = g.tw / 4;
    g.ppix = (g.th + 2) * g.pw;
    g.ppieces = (g.ppix + 15) / 16;
    g.tiles_x = cdiv(a.wo, g.tw);
    g.tiles_y = cdiv(a.ho, g.th);
    auto magic = [](int dv) { const uint64_t v = (((uint64_t)1 << 32) / (uint64_t)dv) + 1u; return (unsigned)(v > 0xffffffffull ? 0xffffffffull : v); };
    g.magic_tw = magic(g.tw);
    g.magic_pw = magic(g.pw);
    a.nblk_m = a.n * g.tiles_x * g.tiles_y;
    a.nblk_n = BN == 96 ? cdiv(a.cout, BN) : cdiv(a.cout_pad, BN);   // (96: the packed rows are padded to a multiple of 128, not of 96 -- no block past cout)
    size_t lds = (size_t)(a.cin > 32 ? 2 : 1) * g.ppieces * 1024 + (size_t)3 * 3 * BN * 64;
    auto kfn = conv_halo8_kernel<DT, ODT, BN, WAVES_M>;
    if (lds < lds_floor_bytes()) lds = lds_floor_bytes();
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    if (getenv("YOLORT_AMD_DEBUG_OCC")) {   // tuning aid
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, 512, lds);
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, (const void*)kfn);
        fprintf(stderr, "[halo8 BN=%d WM=%d] patch %dx%d lds %zu B grid %d: occupancy %d blocks/CU (err %d), regs %d, static lds %zu, max dyn %d\n", BN, WAVES_M, g.th, g.tw, lds,
                a.nblk_m * a.nblk_n, nb, (int)e, fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
    }
    hipLaunchKernelGGL(kfn, dim3(a.nblk_m * a.nblk_n), dim3(512), lds, s, a, g);
    return check_launch("conv_halo8_kernel");
}

template <int DT, int ODT>
static int halo8_variant(const ConvArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        case 1: return launch_halo8<DT, ODT, 128, 4>(a, s);   // 4x2 waves of 64 px x 64 cout
        case 2: return launch_halo8<DT, ODT, 64, 4>(a, s);    // 4x2 waves of 64 px x 32 cout
        case 3: return launch_halo8<DT, ODT, 64, 8>(a, s);    // 8x1 waves of 32 px x 64 cout
        case 4: return launch_halo8<DT, ODT, 32, 8>(a, s);    // 8x1 waves of 32 px x 32 cout
        case 5: return launch_halo8<DT, ODT, 128, 8>(a, s);   // 8x1 waves of 32 px x 128 cout
        case 6: return launch_halo8<DT, ODT, 96, 8>(a, s);    // 8x1 waves of 32 px x 96 cout (round 4: yolov5m's 96 / 192-cout layers paid for a quarter of idle MFMAs in the 128-wide blocks)
        default: set_error("ymi_conv2d: unknown halo8 variant %d", variant); return YMI_EINVAL;
    }
}

int conv_halo8_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.sh == 1 && a.sw == 1 && a.ph == 1 && a.pw == 1 && a.cin % 32 == 0 && a.zeros != nullptr && a.split == 0 && a.up2 == 0,
                "ymi_conv2d: the 8-wave LDS-halo kernel handles plain 3x3 stride-1 pad-1 convolutions with cin %% 32 == 0 (and needs desc.zeros)");
    YMI_REQUIRE(variant != 6 || (a.cout % 96 == 0 && a.chain_w == nullptr), "ymi_conv2d: tile 96 (96-cout blocks) needs cout %% 96 == 0 and no chained convolution");
    if (a.chain_w != nullptr) {   // chained 1x1: the 8 x 1 variants only, one cout block whose width is the chain's fresh K
        const int bn = variant == 3 ? 64 : (variant == 4 ? 32 : (variant == 5 ? 128 : 0));
        YMI_REQUIRE(bn != 0 && bn == a.chain_k && a.cout_pad == bn && out_dtype == dtype,
                    "ymi_conv2d: this halo8 variant does not fit the chained convolution (cout width must equal %d)", a.chain_k);
    }
    YMI_REQUIRE(a.k_pad == 9 * a.cin, "ymi_conv2d: halo8 kernel expects k_pad == 9*cin");
    if (dtype == YMI_F16) return out_dtype == YMI_F32 ? halo8_variant<YMI_F16, YMI_F32>(a, variant, s) : halo8_variant<YMI_F16, YMI_F16>(a, variant, s);
    return out_dtype == YMI_F32 ? halo8_variant<YMI_BF16, YMI_F32>(a, variant, s) : halo8_variant<YMI_BF16, YMI_BF16>(a, variant, s);
}

}  // namespace ymi

#ifdef YMI_STAMPS
extern "C" int ymi_debug_stamps_h8(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ymi::ymi_stamps_h8), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif
