// 3x3 convolution of a 32-channel input (stride 1 or 2, pad 1, cout = 32 or 64), gfx950: the whole weight matrix RESIDENT in
// LDS, persistent blocks walking 8 x 16 output tiles.
//
// These are the two largest-map 3x3 layers of yolov5s (body.1: 32 -> 64 stride 2 at 320^2 -> 160^2, body.2.m.0.cv2:
// 32 -> 32 at 160^2) -- 3200 / 6400 tiles per 32-image batch, each with 18 / 36 MFMAs of work per wave.  The tiled kernels
// give every tile its own block: geometry set-up (~400 instructions), a cold DMA round trip, a weight stage per kernel row
// re-fetched from L2 by every block (18-36 KiB against an 11-35 KiB patch) and a barrier per row -- 47 us and 109 us against
// HBM bounds of 17 and 50 us (profiles/r02m_layer_table_c2.csv).  Here a 4-wave block loads the folded weights ONCE, in
// MFMA fragment order (a wave's weight read is one conflict-free 1 KiB sweep, no swizzle arithmetic), keeps the per-lane
// fragment addresses of the nine taps across tiles (the tile geometry never changes), and per tile only: DMAs the
// (8S+3-S) x (16S+3-S) input patch, waits, runs 18 x TN MFMAs per wave straight through, stores.  Latency is hidden by
// co-resident blocks (5 per CU at stride 1, 2 at stride 2), not by a ring.
//
// Stride 2 keeps the patch columns split by parity ([even columns | odd columns] per row): the 32 lanes of a fragment read
// then touch CONSECUTIVE 64-byte pixels like the stride-1 case (a plain row-major patch has a 2-way bank conflict on every
// read: 16 lanes x 128-byte stride cover 8 of the 16 bank groups).  The permutation costs nothing: LDS-DMA destinations are
// lane-linear anyway, each lane just fetches a different source pixel.
//
// Same arithmetic, accumulator layout and epilogue as the other conv kernels (conv_common.hpp): K order (ky, kx, c), fp32
// accumulate on top of the bias, SiLU (+ residual), channel-slice views.
// Replaces yolort/v5/models/common.py:69-70,115-116 for Conv(32, 64, 3, 2) and Bottleneck(32, 32).cv2.
#include "conv_common.hpp"

namespace ymi {

constexpr int C32_TH = 8, C32_TW = 16;   // output tile: 128 pixels, one 32-pixel group (2 rows) per wave

template <int DT, int S, int TN>
__global__ __launch_bounds__(256, (S == 1 && TN == 1) ? 4 : 2) void conv3x3_c32_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles) {
    typedef typename Mfma<DT>::frag frag;
    constexpr int PH = C32_TH * S + 3 - S, PW = C32_TW * S + 3 - S;   // patch: 10 x 18 (stride 1), 17 x 33 (stride 2)
    constexpr int NE = S == 1 ? PW : C32_TW + 1;                      // stride 2: even columns first (17 of them), then the 16 odd ones
    constexpr int PPIX = PH * PW;
    constexpr int PPIECES = (PPIX + 15) / 16;                         // 12 / 36 DMA pieces of 16 pixels x 64 B
    constexpr int PPW = (PPIECES + 3) / 4;                            // pieces per wave: 3 / 9
    extern __shared__ __attribute__((aligned(16))) unsigned char c32_sm[];
    frag* wl = reinterpret_cast<frag*>(c32_sm);                                 // [(tap*2 + ks)*TN + i][64 lanes] x 16 B
    f32x4* bl = reinterpret_cast<f32x4*>(c32_sm + 18 * TN * 1024);              // [TN][4 groups][2 halves]
    uint16_t* patch = reinterpret_cast<uint16_t*>(c32_sm + 18 * TN * 1024 + TN * 8 * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;

    // ---- resident weights: fragment (tap, ks, i) = rows i*32 + frow, k = tap*32 + ks*16 + hi*8 .. +7 ----
    for (int f = wave; f < 18 * TN; f += 4) {
        const int i = f % TN, ts = f / TN;
        wl[f * 64 + lane] = *reinterpret_cast<const frag*>(a.w + (int64_t)(i * 32 + frow) * a.k_pad + ts * 16 + hi * 8);
    }
    for (int i = tid; i < TN * 8; i += 256) {   // bias quad of (tile t, group g, half h): couts t*32 + g*8 + h*4 ..
        const int t = i >> 3, g = (i >> 1) & 3, h = i & 1;
        bl[i] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + g * 8 + h * 4);
    }

    // ---- patch DMA geometry (fixed per lane): piece pi = 16 patch slots; lane (slot q = pi*16 + lane/4, position lane & 3)
    //      fetches k-chunk pos ^ ((q >> 2) & 3) of the slot's source pixel (pr, pc) relative to the patch origin ----
    int p_rc[PPW];     // pr << 16 | pc, or -1 past the patch
    int p_kc[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        int pi = wave * PPW + j;
        pi = pi < PPIECES ? pi : PPIECES - 1;              // surplus slots re-send the last piece (identical bytes)
        const int q = pi * 16 + (lane >> 2);
        const int qc = q < PPIX ? q : PPIX - 1;
        const int pr = qc / PW, rem = qc - pr * PW;
        const int pc = S == 1 ? rem : (rem < NE ? 2 * rem : 2 * (rem - NE) + 1);
        p_rc[j] = q < PPIX ? ((pr << 16) | pc) : -1;
        p_kc[j] = ((lane & 3) ^ ((q >> 2) & 3)) * 8;
    }
    // ---- fragment geometry (fixed per lane): output pixel p = wave*32 + frow -> (r, c); tap (dy, dx) reads patch slot
    //      (S*r + dy) * PW + col(S*c + dx) ----
    const int pr_o = (wave * 32 + frow) / C32_TW, pc_o = (wave * 32 + frow) % C32_TW;
    int ea[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        const int col = S * pc_o + dx;
        const int slot = S == 1 ? col : ((col & 1) ? NE + (col >> 1) : (col >> 1));
        const int q = (S * pr_o + dy) * PW + slot;
        ea[t] = (q * 32 + ((hi ^ ((q >> 2) & 3)) * 8)) * 2;
    }
    const unsigned char* const pb = reinterpret_cast<const unsigned char*>(patch);

    for (int idx = blockIdx.x; idx < ntiles; idx += gridDim.x) {
        int t = xcd_remap(idx, ntiles);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        const int img = t / tiles_y;
        const int oy0 = ty * C32_TH, ox0 = tx * C32_TW;
        __syncthreads();   // everyone is done reading the previous tile's patch (first pass: the resident weights are written)
        const int iy0 = S * oy0 - 1, ix0 = S * ox0 - 1;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int iy = iy0 + (p_rc[j] >> 16), ix = ix0 + (p_rc[j] & 0xffff);
            const bool ok = p_rc[j] >= 0 && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
            const int off = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + p_kc[j] : a.x_zero_off;
            int pi = wave * PPW + j;
            pi = pi < PPIECES ? pi : PPIECES - 1;
            glds16(a.x + off, patch + pi * 512);
        }
        f32x16 acc[TN][1];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = bl[(i * 4 + g) * 2 + hi];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][0][g * 4 + e] = b[e];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the whole patch has landed
#pragma unroll
        for (int ts = 0; ts < 18; ++ts) {   // (tap, k16 half)
            const frag fa = *reinterpret_cast<const frag*>(pb + ((ts & 1) ? (ea[ts >> 1] ^ 32) : ea[ts >> 1]));
#pragma unroll
            for (int i = 0; i < TN; ++i) acc[i][0] = Mfma<DT>::run(wl[(ts * TN + i) * 64 + lane], fa, acc[i][0]);
        }
        finish_wave_tile<DT, DT, TN, 1>(a, acc, 0, hi, [&](int, int64_t& m, bool& ok) {
            const int oy = oy0 + pr_o, ox = ox0 + pc_o;
            ok = oy < a.ho && ox < a.wo;
            m = ((int64_t)img * a.ho + oy) * a.wo + ox;
        });
    }
}

template <int DT, int S, int TN>
static int launch_c32(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    constexpr int PH = C32_TH * S + 3 - S, PW = C32_TW * S + 3 - S;
    constexpr int PPIECES = (PH * PW + 15) / 16;
    const int tiles_x = cdiv(a.wo, C32_TW), tiles_y = cdiv(a.ho, C32_TH);
    const int ntiles = a.n * tiles_x * tiles_y;
    const size_t lds = (size_t)18 * TN * 1024 + (size_t)TN * 8 * 16 + (size_t)PPIECES * 1024;
    auto kfn = conv3x3_c32_kernel<DT, S, TN>;
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    // persistent blocks: as many as are resident at once (a multiple of 8 so that a block stays on one XCD's tile range)
    static int per_cu = 0;
    if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, lds) != hipSuccess || per_cu < 1)) per_cu = 1;
    const int resident = per_cu * 256;   // MI355X: 256 CUs
    a.nblk_m = ntiles;
    a.nblk_n = 1;
    hipLaunchKernelGGL(kfn, dim3(ntiles < resident ? ntiles : resident), dim3(256), lds, s, a, tiles_x, tiles_y, ntiles);
    return check_launch("conv3x3_c32_kernel");
}

// variant 1 (the only one): stride and cout select the instantiation
int conv3x3_c32_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(variant == 1, "ymi_conv2d: unknown c32 variant %d", variant);
    YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.ph == 1 && a.pw == 1 && a.sh == a.sw && (a.sh == 1 || a.sh == 2) && a.cin == 32 && a.k_pad == 288 &&
                    (a.cout == 32 || a.cout == 64) && a.cout_pad >= a.cout && a.zeros != nullptr && a.up2 == 0 && a.chain_w == nullptr && out_dtype == dtype,
                "ymi_conv2d: the resident-weights 3x3 kernel handles cin = 32, cout = 32 / 64, stride 1 / 2, pad 1, 16-bit output (and needs desc.zeros)");
    const bool f16 = dtype == YMI_F16;
    if (a.sh == 1) {
        if (a.cout == 32) return f16 ? launch_c32<YMI_F16, 1, 1>(a, s) : launch_c32<YMI_BF16, 1, 1>(a, s);
        return f16 ? launch_c32<YMI_F16, 1, 2>(a, s) : launch_c32<YMI_BF16, 1, 2>(a, s);
    }
    if (a.cout == 32) return f16 ? launch_c32<YMI_F16, 2, 1>(a, s) : launch_c32<YMI_BF16, 2, 1>(a, s);
    return f16 ? launch_c32<YMI_F16, 2, 2>(a, s) : launch_c32<YMI_BF16, 2, 2>(a, s);
}

}  // namespace ymi
