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

#ifdef YMI_STAMPS   // tuning aid (never in the shipped build): s_memtime timeline of wave 0 of each block, 8 slots per tile (tools/stamp_r3.py)
__device__ unsigned long long ymi_stamps_r3[256 * 128];
#define R3_STAMP(tile, slot)                                                                                                                        \
    do {                                                                                                                                            \
        if (threadIdx.x == 0 && (tile) < 16) ymi_stamps_r3[blockIdx.x * 128 + (tile) * 8 + (slot)] = __builtin_readcyclecounter();                   \
    } while (0)
#else
#define R3_STAMP(tile, slot) ((void)0)
#endif

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
    int p_rc[R3_PPW];     // pr << 16 | chunk << 8 | pc, or -1: nothing to fetch (past the patch, or one of the two unused chunks of a 48-channel pixel)
    int p_off[R3_PPW];
#pragma unroll
    for (int j = 0; j < R3_PPW; ++j) {
        int pi = wave * R3_PPW + j;
        pi = pi < R3_PIECES ? pi : R3_PIECES - 1;          // surplus slots re-send the last piece (identical bytes)
        const int e = pi * 64 + lane;
        const int q = e >> 3;
        const int qc = q < R3_SLOTS ? q : R3_SLOTS - 1;
        const int pr = qc / R3_PITCH, pc = qc - pr * R3_PITCH;
        const int chunk = (e & 7) ^ (((pr * R3_T + pc) >> 1) & 7);
        p_rc[j] = (q < R3_SLOTS && chunk < NCH) ? ((pr << 16) | (chunk << 8) | pc) : -1;
        p_off[j] = (pr * a.w_in + pc) * a.x_cs + chunk * 8;   // relative to the patch origin: all an interior tile needs
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
    // one DMA piece of the patch of the tile at (img, oy0, ox0).  Interior tiles (the patch lies inside the image: most of them) take the offsets precomputed above
    auto issue_piece = [&](auto jt, int img, int oy0, int ox0, unsigned char* dst) {
        constexpr int j = decltype(jt)::value;
        int pi = wave * R3_PPW + j;
        pi = pi < R3_PIECES ? pi : R3_PIECES - 1;
        int off;
        if (oy0 >= 1 && ox0 >= 1 && oy0 + R3_T + 1 <= a.h && ox0 + R3_T + 1 <= a.w_in) {   // wave-uniform
            off = p_rc[j] >= 0 ? ((img * a.h + oy0 - 1) * a.w_in + ox0 - 1) * a.x_cs + p_off[j] : a.x_zero_off;
        } else {
            const int iy = oy0 - 1 + (p_rc[j] >> 16), ix = ox0 - 1 + (p_rc[j] & 0xff);
            const bool ok = p_rc[j] >= 0 && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
            off = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + ((p_rc[j] >> 8) & 0xff) * 8 : a.x_zero_off;
        }
        glds16(a.x + off, reinterpret_cast<uint16_t*>(dst + pi * 1024));
    };
    auto issue_patch = [&](int idx, unsigned char* dst) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        static_for<0, R3_PPW>([&](auto jt) { issue_piece(jt, img, oy0, ox0, dst); });
    };

    // Lean case (SiLU, no chained conv): the epilogue of a tile is DEFERRED into the next tile's MFMA loop.  Ablation of the first version of this kernel, whose
    // waves ran their 72 MFMAs and then their 32 SiLUs per lane (64 -> 64 at 320^2, bs 8, 12.5 tiles per block, profiles/r03z4_res3x3_ablation.txt): whole kernel 92 us;
    // without the MFMA loop 49; without the epilogue 67; without either 29 (the bare DMA pipeline, one 41 KiB patch in flight per CU: 3.6 TB/s); weight / activation
    // fragments read once instead of per step 90 / 88, both 82 -- the phases of a tile simply ADD (MFMAs 2.6 us, SiLUs 2.0 us, sync + exposed DMA 1.2 - 2.3 us), because the
    // 8 waves meet at the barrier and are then all in the same phase, using the same pipe; LDS fragment traffic costs 11 us of the 92.  Deferring the stores alone (past the
    // next patch's DMA issue) and prefetching the shortcut changed nothing: the round trips were not what bounds a tile.
    // (cout may end 16 channels short of the last 32-wide group -- yolov5m's 48: its second packet pair is then neither fetched nor stored)
    const int64_t cs_max = a.y_cs > a.res_cs ? a.y_cs : a.res_cs;
    const bool lean = !CHAIN && a.act == YMI_ACT_SILU && (a.cout & 15) == 0 && a.cout > 32 * (TN - 1) && ((int64_t)a.M + 1) * cs_max < ((int64_t)1 << 31);
    const bool has_res = a.res != nullptr;
    // the shortcut in packet form (conv_common.hpp, lean_load_residual): two 16-byte loads per 32-channel group, unswapped where they are used
    auto load_res = [&](const LeanPix& p, u32x4 (&rw)[TN][2]) {
        const char* const rb = reinterpret_cast<const char*>(a.res);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (i * 32 + q * 16 < a.cout) rw[i][q] = *reinterpret_cast<const u32x4*>(rb + (size_t)p.ro + (size_t)(8 * hi) + (i * 32 + q * 16) * 2);
    };
    // The epilogue of tile i-1 runs INSIDE the MFMA loop of tile i, one pair of outputs per step: step s = (sub-tile i, octet pair g, half h, pair p) rounds two
    // accumulators of the previous tile exactly like silu_pack_subtile (same helper, same order of operations); every fourth step completes a 16-byte packet
    // (the lane swap) and stores it.  A wave's vector-ALU work then sits between its own MFMAs instead of after them.
    constexpr int NS = 8 * TN;
    f32x16 acc_prev[TN];
    u32x4 rw[TN][2] = {};
    u32x2 rv[4] = {};   // the current packet's four 4-channel pieces (rows g, g + 1 of the group)
    uint32_t pk[2][2] = {};
    LeanPix p_prev;
    p_prev.ok = false;
    bool have_prev = false;
    auto epi_step = [&](auto st) {
        constexpr int s = decltype(st)::value;
        constexpr int i = s >> 3, g = ((s >> 2) & 1) * 2, h = (s >> 1) & 1, p = s & 1;
        if constexpr ((s & 3) == 0) {
            if (has_res) unswap_residual_packet(rw[i][g >> 1], rv, g);
        }
        f32x2 v = {acc_prev[i][(g + h) * 4 + 2 * p], acc_prev[i][(g + h) * 4 + 2 * p + 1]};
        v = silu_pair(v);
        if (has_res) v = v + unpack16<DT>(rv[g + h][p]);
        pk[h][p] = cvt_pk16<DT>(v);
        if constexpr ((s & 3) == 3) {
            const auto rx = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            const u32x4 q = {rx[0], ry[0], rx[1], ry[1]};
            constexpr int co = i * 32 + (g >> 1) * 16;
            if (p_prev.ok && co < a.cout) st16(reinterpret_cast<char*>(a.y) + (size_t)co * 2 + (size_t)p_prev.yo, q);
        }
    };

    int idx = blockIdx.x;
    int buf = 0;
    if (idx < ntiles) issue_patch(idx, patch0);
    int tile_no = 0;
    (void)tile_no;
    for (; idx < ntiles; idx += gridDim.x, ++tile_no) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        R3_STAMP(tile_no, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        R3_STAMP(tile_no, 1);
        __syncthreads();   // patch i has landed; everyone is done reading patch i-1 (first pass: the weights are written)
        R3_STAMP(tile_no, 2);
        const unsigned char* pb = patch0 + buf * R3_PATCH_BYTES;
        // The previous tile's shortcut FIRST: vmcnt retires in order, so a load issued behind the next patch's DMA is only known to have landed when that whole patch
        // has (timeline of the first version, 64 -> 64 at 320^2 with a shortcut: 2035 cycles of a 14 100-cycle tile waiting right here, profiles/r03z9_res3x3_timeline.txt)
        if (lean && have_prev && has_res) load_res(p_prev, rw);
        // (issuing the DMA of the next patch piece by piece between the MFMAs below moved its 1300 cycles per tile into the loop, one for one: a wave stalls at a
        // vector-memory instruction until the texture path accepts it, and nothing behind it issues -- profiles/r03z9_res3x3_timeline.txt)
        if (idx + (int)gridDim.x < ntiles) issue_patch(idx + gridDim.x, patch0 + (buf ^ 1) * R3_PATCH_BYTES);
        buf ^= 1;
        R3_STAMP(tile_no, 3);
        auto pix = [&](int, int64_t& m, bool& ok) {
            const int oy = oy0 + pr_o, ox = ox0 + pc_o;
            ok = oy < a.ho && ox < a.wo;
            m = ((int64_t)img * a.ho + oy) * a.wo + ox;
        };
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
        // epilogue steps run in the last two thirds of the loop: the shortcut fetched at the top of this tile is a memory round trip away (with the steps starting at
        // unit 1 the loop waited 2000 cycles for it at its head)
        constexpr int E0 = NU / 3, SPU = (NS + (NU - E0) - 1) / (NU - E0);
        frag fa[2][U], fw[2][U][TN];
        // (the 9 * KC fragment addresses are invariant across tiles: left alone, the compiler hoists all of them out of the tile loop -- 36 registers at CIN = 64,
        // and the kernel spills; laundering the tap's base address inside the loop keeps the one-instruction XORs where they are used)
        auto read_unit = [&](auto ut, auto bt) {
            constexpr int u = decltype(ut)::value, b = decltype(bt)::value;
            constexpr int t = (u * U) / KC, kc0 = (u * U) % KC;
            int eb = ea[t];
            asm volatile("" : "+v"(eb));
#pragma unroll
            for (int k = 0; k < U; ++k) {
                fa[b][k] = *reinterpret_cast<const frag*>(pb + (eb ^ ((kc0 + k) << 5)));
#pragma unroll
                for (int i = 0; i < TN; ++i) fw[b][k][i] = wl[((t * KC + kc0 + k) * TN + i) * 64 + lane];
            }
        };
        auto run_tile = [&](auto with_epi) {
            constexpr bool EPI = decltype(with_epi)::value;
            read_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            static_for<0, NU>([&](auto ut) {
                constexpr int u = decltype(ut)::value;
                if constexpr (u + 1 < NU) read_unit(std::integral_constant<int, u + 1>{}, std::integral_constant<int, (u + 1) & 1>{});
#pragma unroll
                for (int k = 0; k < U; ++k)
#pragma unroll
                    for (int i = 0; i < TN; ++i) acc[i][0] = Mfma<DT>::run(fw[u & 1][k][i], fa[u & 1][k], acc[i][0]);
                if constexpr (EPI && u >= E0) {
                    static_for<0, SPU>([&](auto jt) {
                        constexpr int s = (u - E0) * SPU + decltype(jt)::value;
                        if constexpr (s < NS) epi_step(std::integral_constant<int, s>{});
                    });
                }
            });
        };
        R3_STAMP(tile_no, 4);
        if (lean && have_prev) run_tile(std::true_type{});
        else run_tile(std::false_type{});
        R3_STAMP(tile_no, 5);
        if constexpr (CHAIN) {
            finish_wave_tile_chain<DT, TN, 1>(a, acc, hi, lane, pix);   // a wave owns ALL couts of its 32 pixels: a chained 1x1 runs from registers
        } else {
            if (lean) {   // handed to the next tile's loop (or to the tail below)
#pragma unroll
                for (int i = 0; i < TN; ++i) acc_prev[i] = acc[i][0];
                p_prev = lean_pix(a, 0, hi, pix);
                have_prev = true;
            } else {
                finish_wave_tile<DT, DT, TN, 1>(a, acc, 0, hi, pix);
            }
        }
        R3_STAMP(tile_no, 6);
    }
    if (have_prev) {   // the last tile's epilogue, in one piece
        if (has_res) load_res(p_prev, rw);
        static_for<0, NS>([&](auto st) { epi_step(st); });
    }
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

#ifdef YMI_STAMPS
extern "C" int ymi_debug_stamps_r3(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ymi::ymi_stamps_r3), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif
