// 3x3 stride-1 "same" convolution of a 48- or 64-channel input into <= 64 channels (gfx950): the whole weight matrix RESIDENT in LDS,
// persistent 8-wave blocks walking 16 x 16 output tiles, the input patch of tile i+1 in flight while tile i is computed.
//
// These are the Bottleneck.cv2 convolutions on the LARGEST maps of every model (yolov5s: 64 -> 64 at 80^2, three per step; yolov5l6: 64 -> 64
// at 320^2; yolov5m: 48 -> 48 at 320^2, run as 64 -> 48 over the zero-padded hidden buffer).  Their K is short -- 576 = 36 k16 steps -- so the
// tiled kernel (conv_halo8.hip: one block per tile, a weight stage per kernel row re-fetched from L2, a barrier per row) spends a tile's time in
// its cold prologue, six barriers and the drain: 9.4 us per 256-pixel tile at 320^2 against 1.9 us of MFMA work and 2.4 us of HBM time
// (profiles/r03z_layer_table_c5.csv: 118 us for a 26 us bound).  Here
//   * the folded weights (9 * CIN * 64 * 2 B = 72 KiB at CIN = 64) are loaded ONCE per block, in MFMA fragment order: a wave's weight read is
//     one conflict-free 1 KiB sweep;
//   * a block is 8 waves (one per CU) on a 16 x 16 tile, a wave owning two tile rows x all couts (the epilogue -- and a chained 1x1 -- of
//     conv_halo8's 8 x 1 form); per tile ONE barrier: "patch i has landed, everyone is done with patch i-1", after which the patch of tile i+1
//     is DMA'd into the other buffer and the 36 x TN MFMAs of the tile run straight through, a tap's fragments fetched under the previous tap's
//     MFMAs;
//   * patch layout: 128-byte pixel slots (CIN = 48 leaves two of the eight 16-byte chunks unused), row pitch 18.  A `ds_read_b128` is served
//     in four groups of 16 lanes whose tile-order pixel indices u cover all residues mod 16 (MI355X_MICROARCH.md, LDS), and a 256-byte bank row
//     holds two slots: with an even pitch the slot parity is u & 1, and the chunk swizzle v = (u >> 1) & 7 supplies the other three bits, so
//     every activation read is conflict-free for every tap.
// Same arithmetic, accumulator layout and epilogue as the other conv kernels (conv_common.hpp): K order (ky, kx, c), fp32 accumulate on top
// of the bias, SiLU (+ residual), channel-slice views; results are bit-identical to the implicit-GEMM kernels' (the halo kernel walks 32-channel chunks
// outermost: same sums, another order, the last bit may differ).
// Replaces yolort/v5/models/common.py:69-70,115-116 for Bottleneck(c, c).cv2 with c_ = 48 / 64.
#include "conv_common.hpp"
#include <cstdlib>

namespace ymi {

constexpr int R3_T = 16;                                  // output tile 16 x 16
constexpr int R3_PH = R3_T + 2, R3_PITCH = R3_T + 2;      // patch rows / row pitch in slots
constexpr int R3_SLOTS = R3_PH * R3_PITCH;                // 324 pixel slots of 128 B
constexpr int R3_ENTRIES = R3_SLOTS * 8;                  // 16-byte entries
constexpr int R3_PIECES = (R3_ENTRIES + 63) / 64;         // 41 DMA pieces of 1 KiB
constexpr int R3_PPW = (R3_PIECES + 7) / 8;               // <= 6 pieces per wave
constexpr int R3_PATCH_BYTES = R3_PIECES * 1024;

template <int DT, int CIN, int TN, bool CHAIN>
__global__ __launch_bounds__(512, 1) void conv3x3_res_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles) {
    typedef typename Mfma<DT>::frag frag;
    constexpr int KC = CIN / 16;            // k16 steps per tap
    constexpr int NCH = CIN / 8;            // real 16-byte chunks of a pixel (6 or 8)
    constexpr int NW = 9 * KC * TN;         // weight fragments (1 KiB each)
    static_assert(CIN == 48 || CIN == 64, "128-byte slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char r3_sm[];
    frag* wl = reinterpret_cast<frag*>(r3_sm);                                   // [(tap*KC + kc)*TN + i][64 lanes] x 16 B
    f32x4* bl = reinterpret_cast<f32x4*>(r3_sm + NW * 1024);                     // [TN][4 groups][2 halves]
    unsigned char* patch0 = r3_sm + NW * 1024 + TN * 8 * 16;                     // two patch buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;

    // ---- resident weights: fragment (tap, kc, i) = rows i*32 + frow, k = tap*CIN + kc*16 + hi*8 .. +7 (packed rows are zero padded to 128) ----
    for (int f = wave; f < NW; f += 8) {
        const int i = f % TN, ts = f / TN;
        wl[f * 64 + lane] = *reinterpret_cast<const frag*>(a.w + (int64_t)(i * 32 + frow) * a.k_pad + ts * 16 + hi * 8);
    }
    if (tid < TN * 8) {   // bias quad of (tile t, group g, half h): couts t*32 + g*8 + h*4 ..
        const int t = tid >> 3, g = (tid >> 1) & 3, h = tid & 1;
        bl[tid] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + g * 8 + h * 4);
    }

    // ---- patch DMA geometry (fixed per lane): entry e = piece*64 + lane -> slot e >> 3 = (pr, pc), position e & 7 holds chunk pos ^ v(pr, pc) ----
    int p_rc[R3_PPW];     // pr << 16 | pc, or -1: nothing to fetch (past the patch, or one of the two unused chunks of a 48-channel pixel)
    int p_kc[R3_PPW];
#pragma unroll
    for (int j = 0; j < R3_PPW; ++j) {
        int pi = wave * R3_PPW + j;
        pi = pi < R3_PIECES ? pi : R3_PIECES - 1;          // surplus slots re-send the last piece (identical bytes)
        const int e = pi * 64 + lane;
        const int q = e >> 3;
        const int qc = q < R3_SLOTS ? q : R3_SLOTS - 1;
        const int pr = qc / R3_PITCH, pc = qc - pr * R3_PITCH;
        const int chunk = (e & 7) ^ (((pr * R3_T + pc) >> 1) & 7);
        p_rc[j] = (q < R3_SLOTS && chunk < NCH) ? ((pr << 16) | pc) : -1;
        p_kc[j] = chunk * 8;
    }
    // ---- fragment geometry (fixed per lane): output pixel p = wave*32 + frow -> (r, c); tap (dy, dx) reads slot (r + dy) * 18 + c + dx ----
    const int pr_o = (wave * 32 + frow) / R3_T, pc_o = (wave * 32 + frow) % R3_T;
    int ea[9];            // byte offset of chunk `hi` (k16 step 0) of the tap's pixel; step kc: ^ (kc << 5)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int pr = pr_o + t / 3, pc = pc_o + t % 3;
        const int v = ((pr * R3_T + pc) >> 1) & 7;
        ea[t] = (pr * R3_PITCH + pc) * 128 + ((hi ^ v) * 16);
    }

    auto tile_origin = [&](int idx, int& img, int& oy0, int& ox0) {
        int t = xcd_remap(idx, ntiles);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        img = t / tiles_y;
        oy0 = ty * R3_T;
        ox0 = tx * R3_T;
    };
    auto issue_patch = [&](int idx, unsigned char* dst) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
#pragma unroll
        for (int j = 0; j < R3_PPW; ++j) {
            int pi = wave * R3_PPW + j;
            pi = pi < R3_PIECES ? pi : R3_PIECES - 1;
            const int iy = oy0 - 1 + (p_rc[j] >> 16), ix = ox0 - 1 + (p_rc[j] & 0xffff);
            const bool ok = p_rc[j] >= 0 && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
            const int off = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + p_kc[j] : a.x_zero_off;
            glds16(a.x + off, reinterpret_cast<uint16_t*>(dst + pi * 1024));
        }
    };

    // Epilogue in two halves around the tile boundary (lean case: SiLU, no chained conv): the packets of tile i stay in registers across the barrier and are
    // stored right after the DMA of tile i+1's patch is issued, at the top of tile i+1 -- they have a whole tile to be acknowledged before the next vmcnt(0) --
    // and the shortcut of tile i is fetched at the top of tile i, a whole MFMA phase before it is added.  (Measured: neither round trip was what bounds a tile.
    // Ablation on 64 -> 64 at 320^2, bs 8, 12.5 tiles per block, profiles/r03z4_res3x3_ablation.txt: whole kernel 92 us; without the MFMA loop 49; without the
    // epilogue 67; without either 29 -- the bare DMA pipeline, one 41 KiB patch in flight per CU: 3.6 TB/s; weight / activation fragments read once instead of per
    // step 90 / 88, both 82 -- so the LDS reads cost 11 us, NOT the 1.5 KiB-per-MFMA wall they were suspected to be.  The phases of a tile simply ADD: 8 waves meet at
    // the barrier, run their 72 MFMAs together (2.6 us against 1.9 at the matrix pipe's rate), then their 32 SiLUs per lane together (2.0 us of vector ALU), and the
    // pipe of the other kind idles meanwhile.  What would overlap them is the previous tile's SiLUs issued BETWEEN this tile's MFMAs, or two desynchronised
    // blocks per CU -- neither fits 256 registers / 160 KiB as the kernel stands.)
    // (cout may end 16 channels short of the last 32-wide group -- yolov5m's 48: its second packet pair is then neither fetched nor stored)
    const int64_t cs_max = a.y_cs > a.res_cs ? a.y_cs : a.res_cs;
    const bool lean = !CHAIN && a.act == YMI_ACT_SILU && (a.cout & 15) == 0 && a.cout > 32 * (TN - 1) && ((int64_t)a.M + 1) * cs_max < ((int64_t)1 << 31);
    const bool has_res = a.res != nullptr;
    auto store_prev = [&](const LeanPix& p, const u32x4 (&o)[TN][2]) {
        if (!p.ok) return;
        char* const yb = reinterpret_cast<char*>(a.y);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (i * 32 + q * 16 < a.cout) st16(yb + (size_t)(i * 32 + q * 16) * 2 + (size_t)p.yo, o[i][q]);
    };
    auto load_res = [&](const LeanPix& p, u32x2 (&rv)[TN][4]) {
        const char* const rb = reinterpret_cast<const char*>(a.res);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (i * 32 + g * 8 < a.cout) rv[i][g] = *reinterpret_cast<const u32x2*>(rb + (size_t)p.ro + (i * 32 + g * 8) * 2);
    };
    u32x4 o_prev[TN][2];
    LeanPix p_prev;
    p_prev.ok = false;
    bool have_prev = false;

    int idx = blockIdx.x;
    int buf = 0;
    if (idx < ntiles) issue_patch(idx, patch0);
    for (; idx < ntiles; idx += gridDim.x) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // patch i has landed; everyone is done reading patch i-1 (first pass: the weights are written)
        const unsigned char* pb = patch0 + buf * R3_PATCH_BYTES;
        if (idx + (int)gridDim.x < ntiles) issue_patch(idx + gridDim.x, patch0 + (buf ^ 1) * R3_PATCH_BYTES);
        buf ^= 1;
        auto pix = [&](int, int64_t& m, bool& ok) {
            const int oy = oy0 + pr_o, ox = ox0 + pc_o;
            ok = oy < a.ho && ox < a.wo;
            m = ((int64_t)img * a.ho + oy) * a.wo + ox;
        };
        LeanPix p_cur;
        u32x2 rv[TN][4] = {};
        if (lean) {
            if (have_prev) store_prev(p_prev, o_prev);
            p_cur = lean_pix(a, 0, hi, pix);
            if (has_res) load_res(p_cur, rv);
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
        // unit = U k16 steps of one tap (half a tap at CIN = 64, a whole one at 48): its U activation and U * TN weight fragments are fetched
        // under the previous unit's MFMAs (a whole tap in flight twice over did not fit the 256 registers of a wave: 84 spilled)
        constexpr int U = KC == 4 ? 2 : KC, NU = 9 * KC / U;
        frag fa[2][U], fw[2][U][TN];
        auto read_unit = [&](auto ut, auto bt) {
            constexpr int u = decltype(ut)::value, b = decltype(bt)::value;
            constexpr int t = (u * U) / KC, kc0 = (u * U) % KC;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                fa[b][k] = *reinterpret_cast<const frag*>(pb + (ea[t] ^ ((kc0 + k) << 5)));
#pragma unroll
                for (int i = 0; i < TN; ++i) fw[b][k][i] = wl[((t * KC + kc0 + k) * TN + i) * 64 + lane];
            }
        };
        read_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, NU>([&](auto ut) {
            constexpr int u = decltype(ut)::value;
            if constexpr (u + 1 < NU) read_unit(std::integral_constant<int, u + 1>{}, std::integral_constant<int, (u + 1) & 1>{});
#pragma unroll
            for (int k = 0; k < U; ++k)
#pragma unroll
                for (int i = 0; i < TN; ++i) acc[i][0] = Mfma<DT>::run(fw[u & 1][k][i], fa[u & 1][k], acc[i][0]);
        });
        if constexpr (CHAIN) {
            finish_wave_tile_chain<DT, TN, 1>(a, acc, hi, lane, pix);   // a wave owns ALL couts of its 32 pixels: a chained 1x1 runs from registers
        } else {
            if (lean) {   // the arithmetic of finish_wave_tile_lean; the stores follow at the top of the next tile
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    if (has_res) silu_pack_subtile<DT, true>(acc[i][0], rv[i], o_prev[i]);
                    else silu_pack_subtile<DT, false>(acc[i][0], rv[i], o_prev[i]);
                }
                p_prev = p_cur;
                have_prev = true;
            } else {
                finish_wave_tile<DT, DT, TN, 1>(a, acc, 0, hi, pix);
            }
        }
    }
    if (have_prev) store_prev(p_prev, o_prev);
}

template <int DT, int CIN, int TN, bool CHAIN>
static int launch_res_t(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int tiles_x = cdiv(a.wo, R3_T), tiles_y = cdiv(a.ho, R3_T);
    const int ntiles = a.n * tiles_x * tiles_y;
    const size_t lds = (size_t)9 * (CIN / 16) * TN * 1024 + (size_t)TN * 8 * 16 + (size_t)2 * R3_PATCH_BYTES;
    auto kfn = conv3x3_res_kernel<DT, CIN, TN, CHAIN>;
    { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    int resident = 256;   // one 8-wave block per CU
    if (const char* e = getenv("YOLORT_AMD_RES3X3_BLOCKS")) {   // test aid: few blocks walk many tiles (the persistent loop on small inputs)
        const int v = atoi(e);
        if (v >= 1 && v <= 256) resident = v;
    }
    a.nblk_m = ntiles;
    a.nblk_n = 1;
    hipLaunchKernelGGL(kfn, dim3(ntiles < resident ? ntiles : resident), dim3(512), lds, s, a, tiles_x, tiles_y, ntiles);
    return check_launch("conv3x3_res_kernel");
}

template <int DT, int CIN, int TN>
static int launch_res(const ConvArgs& a, hipStream_t s) {
    return a.chain_w != nullptr ? launch_res_t<DT, CIN, TN, true>(a, s) : launch_res_t<DT, CIN, TN, false>(a, s);
}

// variant 1 (the only one): cin and cout select the instantiation
int conv3x3_res_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(variant == 1, "ymi_conv2d: unknown resident-weights 3x3 variant %d", variant);
    YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.ph == 1 && a.pw == 1 && a.sh == 1 && a.sw == 1 && (a.cin == 48 || a.cin == 64) && a.k_pad >= 9 * a.cin &&
                    a.cout >= 1 && a.cout <= 64 && a.cout_pad >= a.cout && a.zeros != nullptr && a.up2 == 0 && a.split == 0 && out_dtype == dtype,
                "ymi_conv2d: the resident-weights 3x3 kernel (tile 132) handles cin = 48 / 64, cout <= 64, stride 1, pad 1, 16-bit output (and needs desc.zeros)");
    YMI_REQUIRE((int64_t)a.n * a.h * a.w_in * a.x_cs < ((int64_t)1 << 31), "ymi_conv2d: input tensor too large for 32-bit offsets");
    if (a.chain_w != nullptr) YMI_REQUIRE(a.cout_pad == a.chain_k && (a.chain_k == 32 || a.chain_k == 64), "ymi_conv2d: tile 132 does not fit the chained convolution (cout width must equal %d)", a.chain_k);
    const bool f16 = dtype == YMI_F16;
    const int tn = a.cout_pad <= 32 ? 1 : 2;
    if (a.cin == 64) {
        if (tn == 1) return f16 ? launch_res<YMI_F16, 64, 1>(a, s) : launch_res<YMI_BF16, 64, 1>(a, s);
        return f16 ? launch_res<YMI_F16, 64, 2>(a, s) : launch_res<YMI_BF16, 64, 2>(a, s);
    }
    if (tn == 1) return f16 ? launch_res<YMI_F16, 48, 1>(a, s) : launch_res<YMI_BF16, 48, 1>(a, s);
    return f16 ? launch_res<YMI_F16, 48, 2>(a, s) : launch_res<YMI_BF16, 48, 2>(a, s);
}

}  // namespace ymi
