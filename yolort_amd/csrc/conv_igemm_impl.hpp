// Kernel templates of the implicit-GEMM convolution family and of the fused detection head (see conv_igemm.hip for the
// design notes).  Included by conv_igemm.hip (dispatch only) and by the conv_inst_*.hip / head_inst_*.hip translation
// units, each of which explicitly instantiates one slice of the tile x dtype space.
#pragma once
#include <string.h>

#include "conv_common.hpp"
#include "head_decode.hpp"

namespace ymi {

// BM x BN block tile, each wave WM x WN; IS1X1: kh=kw=1, stride 1, pad 0 (no bounds checks, no table)
template <int DT, int ODT, int BM, int BN, int WM, int WN, bool IS1X1>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_ROWS = BM / 64, W_ROWS = (BN + 63) / 64;  // rows per thread per tile
    constexpr int WAVES_N = BN / WN;
    typedef typename Mfma<DT>::frag frag;

    __shared__ __attribute__((aligned(16))) uint16_t lds[2][(BM + BN) * LDS_PITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = (wave / WAVES_N) * WM, wave_n = (wave % WAVES_N) * WN;

    // block -> tile: XCD-aware remap, then cout-tile fastest so the blocks that share one
    // activation tile run back-to-back on the same XCD (its L2 serves the re-reads).
    const int nblk = a.nblk_m * a.nblk_n;
    const int lb = xcd_remap(blockIdx.x, nblk);
    const int bm = lb / a.nblk_n, bn = lb % a.nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- per-thread gather geometry: thread loads chunk (tid&3) of rows (tid>>2) + 64*i ----
    const int chunk = tid & 3;
    const int row0 = tid >> 2;
    int64_t a_base[A_ROWS];   // element offset of (img, iy0, ix0, 0); may be "negative-ish" -> int64
    int a_iy0[A_ROWS], a_ix0[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + row0 + 64 * i;
        if (m < a.M) {
            const int img = m / (a.ho * a.wo);
            const int rem = m - img * (a.ho * a.wo);
            const int oy = rem / a.wo, ox = rem - oy * a.wo;
            const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
            a_iy0[i] = iy0;
            a_ix0[i] = ix0;
            a_base[i] = ((int64_t)(img * a.h + iy0) * a.w_in + ix0) * a.x_cs;
        } else {
            a_iy0[i] = -100000;  // every tap out of range -> zeros
            a_ix0[i] = -100000;
            a_base[i] = 0;
        }
    }
    const uint16_t* w_ptr[W_ROWS];
    bool w_ok[W_ROWS];
#pragma unroll
    for (int i = 0; i < W_ROWS; ++i) {
        const int r = row0 + 64 * i;
        w_ok[i] = (r < BN) && (n0 + r < a.cout_pad);
        w_ptr[i] = a.w + (int64_t)(n0 + (w_ok[i] ? r : 0)) * a.k_pad + chunk * 8;
    }

    u32x4 a_reg[A_ROWS], w_reg[W_ROWS];
    const int nsteps = a.k_pad / BK;

    auto load_tiles = [&](int step) {
        const int q = step * 4 + chunk;  // 8-channel chunk index along K
        int koff, dy, dx;
        if constexpr (IS1X1) {
            koff = q * 8;
            dy = 0;
            dx = 0;
        } else {
            const int2 t = a.ktab[q];
            koff = t.x;
            dy = t.y >> 16;      // -1 for padding chunks (t.y == -1)
            dx = t.y & 0xffff;
        }
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            bool ok;
            if constexpr (IS1X1) {
                ok = (a_iy0[i] >= 0) && (koff < a.cin);
            } else {
                const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
                ok = (dy >= 0) && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
            }
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(a.x + a_base[i] + koff);
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_ROWS; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (w_ok[i]) v = *reinterpret_cast<const u32x4*>(w_ptr[i] + step * BK);
            w_reg[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        uint16_t* as = lds[buf];
        uint16_t* ws = lds[buf] + BM * LDS_PITCH;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            *reinterpret_cast<u32x4*>(as + (row0 + 64 * i) * LDS_PITCH + chunk * 8) = a_reg[i];
#pragma unroll
        for (int i = 0; i < W_ROWS; ++i)
            if (row0 + 64 * i < BN) *reinterpret_cast<u32x4*>(ws + (row0 + 64 * i) * LDS_PITCH + chunk * 8) = w_reg[i];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) load_tiles(step + 1);  // global loads in flight under the MFMAs
        const uint16_t* as = lds[buf] + wave_m * LDS_PITCH;
        const uint16_t* ws = lds[buf] + (BM + wave_n) * LDS_PITCH;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag af[TM], wf[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j)
                af[j] = *reinterpret_cast<const frag*>(as + (j * 32 + frow) * LDS_PITCH + ks * 16 + fk);
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *reinterpret_cast<const frag*>(ws + (i * 32 + frow) * LDS_PITCH + ks * 16 + fk);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = Mfma<DT>::run(wf[i], af[j], acc[i][j]);
        }
        if (step + 1 < nsteps) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + act (+ residual), lane owns pixel (lane&31) and 4 groups of 4 couts ----
    const int hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wave_m + j * 32 + frow;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = n0 + wave_n + i * 32 + g * 8 + hi * 4;
                if (co >= a.cout) continue;
                const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][g * 4 + e] + b[e];
                    if (a.act == YMI_ACT_SILU) t = silu(t);
                    v[e] = t;
                }
                if (a.res != nullptr) {
                    const u32x2 rv = *reinterpret_cast<const u32x2*>(a.res + (int64_t)m * a.res_cs + co);
                    v[0] += from16<DT>((uint16_t)(rv[0] & 0xffff));
                    v[1] += from16<DT>((uint16_t)(rv[0] >> 16));
                    v[2] += from16<DT>((uint16_t)(rv[1] & 0xffff));
                    v[3] += from16<DT>((uint16_t)(rv[1] >> 16));
                }
                if constexpr (ODT == YMI_F32) {
                    float* yp = reinterpret_cast<float*>(a.y) + (int64_t)m * a.y_cs + co;
                    if (co + 3 < a.cout) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(yp) = o;
                    } else {
                        for (int e = 0; e < 4 && co + e < a.cout; ++e) yp[e] = v[e];
                    }
                } else {
                    uint16_t* yp = reinterpret_cast<uint16_t*>(a.y) + (int64_t)m * a.y_cs + co;
                    if (co + 3 < a.cout) {
                        u32x2 o;
                        o[0] = cvt_pk16<DT>(f32x2{v[0], v[1]});
                        o[1] = cvt_pk16<DT>(f32x2{v[2], v[3]});
                        *reinterpret_cast<u32x2*>(yp) = o;
                    } else {
                        for (int e = 0; e < 4 && co + e < a.cout; ++e) yp[e] = to16<DT>(v[e]);
                    }
                }
            }
        }
    }
}


// =============================================================================================
// v2: LDS-DMA pipelined variant.  Same tiling and MFMA mapping as above, but operand tiles travel
// HBM -> LDS with `global_load_lds_dwordx4` (no VGPR round trip) into a STAGES-deep ring, so
// STAGES-1 k-steps of loads are in flight behind the MFMAs and there is ONE barrier per k-step
// (cdna_hip_programming.md section 5: counted vmcnt + raw s_barrier; all LDS in one array).
//   * a wave-instruction moves 64 lanes x 16 B = 1 KiB to LDS base + lane*16: 16 tile rows x 64 B.
//     LDS rows are therefore dense (64 B), and bank conflicts of the fragment reads are removed by
//     an XOR swizzle applied on the SOURCE side: lane (row, pos) fetches k-chunk pos ^ ((row>>2)&3),
//     the reader of chunk c looks at position c ^ ((row>>2)&3) (rule 21: linear dest, permuted
//     source, same involution on the read).
//   * out-of-image taps, rows past M / cout_pad and K padding read from a zero page instead of
//     branching, so every lane always issues its load (LDS slots must be overwritten each round).
//   * the im2col table lives in LDS (ds_read, lgkmcnt) so that no ordinary VMEM load sits in the
//     main loop -- hipcc would otherwise drain the DMA queue with vmcnt(0) at its first use.
// Epilogue: bias + SiLU (+ residual) in fp32, then lanes l / l+32 exchange halves with
// v_permlane32_swap so that each lane stores 8 consecutive output channels (16 B) per store.
// =============================================================================================
// UTAP (uniform tap): cin % 32 == 0 and kh*kw <= 32 -> scalar tap arithmetic + per-row validity bitmask, no im2col table
#ifdef YMI_STAMPS   // tuning aid (never in the shipped build): s_memtime timeline of the pipelined main loop, wave 0 of each block
__device__ unsigned long long ymi_stamps[2048 * 128];
#define YMI_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 2048 && (i) < 128) ymi_stamps[blockIdx.x * 128 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define YMI_STAMP(i) ((void)0)
#endif

// PIPE: software-pipelined main loop -- MFMA fragments are double-buffered in registers (the LDS reads of the next
// half-step and the DMA issue of a later stage sit between the MFMAs of the current one), so a single wave keeps
// its SIMD's matrix pipe busy instead of serialising wait -> barrier -> DMA issue -> LDS latency -> MFMA.
// The epilogue is a functor: epi(acc, m0 + wave_m, n0 + wave_n, lane, wave, smem) -- plain stores (StoreEpilogue)
// or the fused detection decode of the head (head_decode.hpp).
template <int DT, int ODT, int BM, int BN, int WM, int WN, int STAGES, bool IS1X1, bool UTAP, bool PIPE, class Epi>
__device__ __forceinline__ void conv_igemm_v2_body(const ConvArgs& a, Epi&& epi, int block_id) {   // block_id: blockIdx.x, or the id within a grouped launch
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert(BM % 64 == 0, "activation pieces are dealt 1:1 to the 4 waves");
    constexpr int PA = BM / 64;                  // activation pieces (1 KiB = 16 rows) per wave per stage
    constexpr int W_PIECES = BN / 16;            // weight pieces per stage (all waves together)
    constexpr int PW = (W_PIECES + 3) / 4;       // weight pieces per wave per stage
    constexpr int P = PA + PW;                   // DMA instructions per wave per stage (same for every wave)
    constexpr int STAGE_HALFS = (BM + BN) * 32;  // uint16 elements per stage
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // the ONLY LDS object
    int2* ktab_lds = reinterpret_cast<int2*>(smem + STAGES * STAGE_HALFS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = (wave / WAVES_N) * WM, wave_n = (wave % WAVES_N) * WN;

    const int nblk = a.nblk_m * a.nblk_n;
    const int lb = xcd_remap(block_id, nblk);
    const int bm = lb / a.nblk_n, bn = lb % a.nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nsteps = (a.debug & 16) ? 1 : a.k_pad / BK;   // bit 4 (tuning aid): first K step only

    if constexpr (!IS1X1 && !UTAP) {
        for (int i = tid; i < a.k_pad / 8; i += 256) ktab_lds[i] = a.ktab[i];
    }

    // ---- per-lane DMA geometry.  Wave w moves activation pieces w*PA .. w*PA+PA-1 (16 rows each) and
    //      weight pieces w*PW .. (clamped: surplus waves re-send the last piece, identical bytes).
    //      Addresses are (uniform 64-bit base) + (per-lane 32-bit element offset); out-of-range
    //      activation chunks select the offset of a zero page that lives in the tail of x's own
    //      buffer, weight rows past cout are real zero rows of the packed tensor -> no branches. ----
    const int sub_row = lane >> 2;                         // row within the piece
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);      // k-chunk fetched = pos ^ ((row>>2)&3)
    int a_off[PA];       // element offset of (img, iy0, ix0, 0) relative to a.x
    int a_aux[PA];       // UTAP: bit t set <=> tap t of this row is inside the image (0 for rows past M)
                         // else: (iy0+16384)<<16 | (ix0+16384), or -1 for rows past M
    int a_slot[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int pi = wave * PA + j;
        a_slot[j] = pi * 512;
        const int m = m0 + pi * 16 + sub_row;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        if constexpr (IS1X1) {   // kh = kw = 1, stride 1, pad 0: the conv reads pixel m itself (no index arithmetic)
            a_off[j] = mm * a.x_cs;
            a_aux[j] = ok ? 0 : -1;   // only the sign is read on this path
            continue;
        }
        const int hw_o = a.ho * a.wo;
        const int img = fast_div(mm, hw_o, a.magic_hw);
        const int rem = mm - img * hw_o;
        const int oy = fast_div(rem, a.wo, a.magic_w), ox = rem - oy * a.wo;
        const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
        a_off[j] = ((img * a.h + iy0) * a.w_in + ix0) * a.x_cs;
        if constexpr (UTAP) {
            unsigned mask = 0;
            int t = 0;
            for (int dy = 0; dy < a.kh; ++dy) {
                const bool yin = (unsigned)(iy0 + dy) < (unsigned)a.h;
                for (int dx = 0; dx < a.kw; ++dx, ++t) {
                    const bool in = yin && ((unsigned)(ix0 + dx) < (unsigned)a.w_in);
                    mask |= (in ? 1u : 0u) << t;
                }
            }
            a_aux[j] = ok ? (int)mask : 0;
        } else {
            a_aux[j] = ok ? (((iy0 + 16384) << 16) | ((ix0 + 16384) & 0xffff)) : -1;
        }
    }
    int w_off[PW];       // element offset of (row, chunk*8) in the packed weights (rows are zero-padded to 128)
    int w_slot[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        int pi = wave * PW + j;
        pi = pi < W_PIECES ? pi : W_PIECES - 1;
        w_slot[j] = (BM / 16 + pi) * 512;
        w_off[j] = (n0 + pi * 16 + sub_row) * a.k_pad + chunk * 8;
    }

    // UTAP running state (stages are issued in order 0,1,2,...): all wave-uniform scalars
    int u_tap = 0, u_c0 = 0, u_dx = 0, u_kbase = 0;
    // one stage = issue_begin(step); issue_piece(0..P-1); issue_end()  (pieces 0..PA-1 activations, PA..P-1 weights)
    uint16_t* cur_stage = smem;
    int cur_koff = 0, cur_step = 0, cur_dy = 0, cur_dx = 0;
    bool cur_tap_ok = true;
    auto issue_begin = [&](int step) {
        cur_stage = smem + (step % STAGES) * STAGE_HALFS;
        cur_step = step;
        if constexpr (UTAP) {
            cur_koff = u_kbase + chunk * 8;   // cin % 32 == 0: the four chunks of a step share one tap -> scalar tap math
        } else if constexpr (IS1X1) {
            cur_koff = (step * 4 + chunk) * 8;
            cur_tap_ok = cur_koff < a.cin;
        } else {
            const int2 t = ktab_lds[step * 4 + chunk];
            cur_koff = t.x;
            cur_tap_ok = t.y >= 0;
            cur_dy = t.y >> 16;
            cur_dx = t.y & 0xffff;
        }
    };
    auto issue_piece = [&](auto jt) {
        constexpr int j = decltype(jt)::value;
        if constexpr (j < PA) {
            bool ok;
            if constexpr (UTAP) {
                ok = (a_aux[j] >> u_tap) & 1;
            } else {
                ok = cur_tap_ok & (a_aux[j] >= 0);
                if constexpr (!IS1X1) {
                    const int iy = (a_aux[j] >> 16) - 16384 + cur_dy, ix = (a_aux[j] & 0xffff) - 16384 + cur_dx;
                    ok = ok & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w_in);
                }
            }
            const int off = ok ? a_off[j] + cur_koff : a.x_zero_off;
            glds16(a.x + off, cur_stage + a_slot[j]);
        } else if constexpr (j < P) {
            glds16(a.w + (w_off[j - PA] + cur_step * BK), cur_stage + w_slot[j - PA]);
        }
    };
    auto issue_end = [&]() {
        if constexpr (UTAP) {
            // advance to the next 32-channel chunk / tap / kernel row (element offsets relative to (iy0, ix0))
            u_c0 += BK;
            u_kbase += BK;
            if (u_c0 == a.cin) {
                u_c0 = 0;
                ++u_tap;
                ++u_dx;
                u_kbase += a.x_cs - a.cin;
                if (u_dx == a.kw) {
                    u_dx = 0;
                    u_kbase += (a.w_in - a.kw) * a.x_cs;
                }
            }
        }
    };
    auto issue = [&](int step) {
        issue_begin(step);
        static_for<0, P>(issue_piece);
        issue_end();
    };

    f32x16 acc[TN][TM];

    if constexpr (!IS1X1 && !UTAP) __syncthreads();   // ktab visible (no DMA in flight yet: plain barrier is fine)
    const int frow = lane & 31;
    const int swz = (lane >> 2) & 3;
    int pos[2];
    pos[0] = ((0 + (lane >> 5)) ^ swz) * 8;   // element offset of this lane's k-chunk, ks = 0
    pos[1] = ((2 + (lane >> 5)) ^ swz) * 8;   // ks = 1

    if constexpr (PIPE) {
        YMI_STAMP(0);
        // ---- software-pipelined main loop: all STAGES slots are in use (one being read, STAGES-1 in flight) ----
        static_for<0, STAGES>([&](auto st) {
            if (decltype(st)::value < nsteps) issue(decltype(st)::value);
        });
        f32x4 bias_regs[TN][4];   // issued behind the prologue DMA (conv_common.hpp)
        load_bias<TN>(a, n0 + wave_n, lane >> 5, bias_regs);
        init_acc<TN, TM>(acc, bias_regs);   // accumulate on top of the bias
        // (after the prologue DMA issue: waiting for the bias load first put two cold memory latencies in series at every block start)
        auto wait_pending = [&](int pend) {   // returns once at most `pend` later stages of this wave are in flight
            if (pend >= 3) wait_vmcnt<3 * P>();
            else if (pend == 2) wait_vmcnt<2 * P>();
            else if (pend == 1) wait_vmcnt<P>();
            else wait_vmcnt<0>();
        };
        YMI_STAMP(1);
        wait_pending((nsteps < STAGES ? nsteps : STAGES) - 1);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        YMI_STAMP(2);

        constexpr int NM = TN * TM;        // MFMAs per half-step
        constexpr int NF = TM + TN;        // fragments per half-step
        frag fa[2][TM], fw[2][TN];
        const uint16_t* as = smem + wave_m * 32;
        const uint16_t* ws = smem + (BM + wave_n) * 32;
        auto read_frag = [&](auto buft, auto qt, const uint16_t* sa, const uint16_t* sw, int p) {
            constexpr int buf = decltype(buft)::value, q = decltype(qt)::value;
            if constexpr (q < TM) fa[buf][q] = *reinterpret_cast<const frag*>(sa + (q * 32 + frow) * 32 + p);
            else if constexpr (q < NF) fw[buf][q - TM] = *reinterpret_cast<const frag*>(sw + ((q - TM) * 32 + frow) * 32 + p);
        };
        // MFMAs on fragment buffer `cur`; after the q-th MFMA run the extra items [q*PER, (q+1)*PER) of `extra`
        auto mfma_group = [&](auto curt, auto nextra_t, auto&& extra) {
            constexpr int cur = decltype(curt)::value, NE = decltype(nextra_t)::value;
            constexpr int PER = (NE + NM - 1) / NM;
            static_for<0, NM>([&](auto qt) {
                constexpr int q = decltype(qt)::value, i = q / TM, j = q % TM;
                acc[i][j] = Mfma<DT>::run(fw[cur][i], fa[cur][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, PER>([&](auto et) {
                    constexpr int e = q * PER + decltype(et)::value;
                    if constexpr (e < NE) extra(std::integral_constant<int, e>{});
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        static_for<0, NF>([&](auto qt) { read_frag(std::integral_constant<int, 0>{}, qt, as, ws, pos[0]); });
        int slot = 0;
        // ONE loop body for every step (no per-case copies of the MFMA groups: the accumulators stay put).  On the last
        // step the "next stage" fragment reads fetch stale LDS bytes that nobody uses, and the wait / barrier are idle.
        for (int step = 0; step < nsteps; ++step) {
            // first half-step; meanwhile fetch the second half-step's fragments of the same stage
            mfma_group(std::integral_constant<int, 0>{}, std::integral_constant<int, NF>{},
                       [&](auto et) { read_frag(std::integral_constant<int, 1>{}, et, as, ws, pos[1]); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave is done reading stage `step`
            const int issued = step + STAGES < nsteps ? step + STAGES : nsteps;
            YMI_STAMP(4 + step * 3);
            wait_pending(issued - (step + 2));                    // this wave's pieces of stage step+1 have landed
            YMI_STAMP(5 + step * 3);
            __builtin_amdgcn_s_barrier();                         // ... everyone's have; slot `slot` is free
            __builtin_amdgcn_sched_barrier(0);
            YMI_STAMP(6 + step * 3);
            const int nslot = slot + 1 == STAGES ? 0 : slot + 1;
            as = smem + nslot * STAGE_HALFS + wave_m * 32;
            ws = smem + nslot * STAGE_HALFS + (BM + wave_n) * 32;
            const bool refill = step + STAGES < nsteps;           // wave-uniform
            if (refill) issue_begin(step + STAGES);
            __builtin_amdgcn_sched_barrier(0);
            // second half-step; meanwhile fetch the next stage's first fragments and refill the freed slot
            mfma_group(std::integral_constant<int, 1>{}, std::integral_constant<int, NF + P>{}, [&](auto et) {
                constexpr int e = decltype(et)::value;
                if constexpr (e < NF) read_frag(std::integral_constant<int, 0>{}, et, as, ws, pos[0]);
                else if (refill) issue_piece(std::integral_constant<int, e - NF>{});
            });
            if (refill) issue_end();
            slot = nslot;
        }
    } else {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps && !(a.debug & 2)) issue(s);
    f32x4 bias_regs[TN][4];   // issued behind the prologue DMA (conv_common.hpp)
    load_bias<TN>(a, n0 + wave_n, lane >> 5, bias_regs);
    init_acc<TN, TM>(acc, bias_regs);   // accumulate on top of the bias

    for (int step = 0; step < nsteps; ++step) {
        // this wave's pieces of stage `step` have landed once at most `ahead` later stages are pending
        const int issued = (step + STAGES - 1 < nsteps) ? step + STAGES - 1 : nsteps;
        const int ahead = issued - (step + 1);
        if (ahead >= 2) wait_vmcnt<2 * P>();
        else if (ahead == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();          // every wave's pieces landed; everyone is done with stage step-1
        __builtin_amdgcn_sched_barrier(0);
        if (step + STAGES - 1 < nsteps && !(a.debug & 2)) issue(step + STAGES - 1);   // refill the slot freed by step-1
        if (a.debug & 1) continue;
        const uint16_t* as = smem + (step % STAGES) * STAGE_HALFS + wave_m * 32;
        const uint16_t* ws = smem + (step % STAGES) * STAGE_HALFS + (BM + wave_n) * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag af[TM], wf[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const frag*>(as + (j * 32 + frow) * 32 + pos[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const frag*>(ws + (i * 32 + frow) * 32 + pos[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = Mfma<DT>::run(wf[i], af[j], acc[i][j]);
        }
    }
    }

    YMI_STAMP(3);
    epi(acc, m0 + wave_m, n0 + wave_n, lane, wave, smem);
    YMI_STAMP(127);
}

// SiLU (+ residual), 16-byte stores straight from the MFMA layout (conv_common.hpp)
template <int DT, int ODT>
struct StoreEpilogue {
    const ConvArgs& a;
    template <int TN, int TM>
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[TN][TM], int mbase, int cbase0, int lane, int, uint16_t*) const {
        auto pix = [&](int j, int64_t& m, bool& ok) {
            m = mbase + j * 32 + (lane & 31);
            ok = m < a.M;
        };
        if constexpr (ODT == DT && TN <= 4 && TN == (TN & -TN)) {   // chained 1x1 (launch checks: this wave tile spans exactly the chain's K channels)
            if (a.chain_w != nullptr && cbase0 == 0) {
                finish_wave_tile_chain<DT, TN, TM>(a, acc, lane >> 5, lane, pix);
                return;
            }
        }
        finish_wave_tile<DT, ODT, TN, TM>(a, acc, cbase0, lane >> 5, pix);
    }
};

// StoreEpilogue with ROW-TRANSPOSED stores (finish_wave_tile_lean_tp, conv_common.hpp): the operand ring is dead once every wave has
// left the main loop -- one block barrier -- and each wave takes 32 * (64 TN + 16) bytes of it.  Waves outside the lean case (ragged
// cout, fp32 output, upsampled copy, a chained 1x1, a split inside the wave's range) store as StoreEpilogue does.
template <int DT, int ODT>
struct StoreEpilogueTP {
    const ConvArgs& a;
    template <int TN, int TM>
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[TN][TM], int mbase, int cbase0, int lane, int wave, uint16_t* smem) const {
        if constexpr (ODT == DT && (TN == 1 || TN == 2 || TN == 4)) {
            __syncthreads();   // every wave of the block is done reading the ring (all of them call the epilogue)
            if (!(a.chain_w != nullptr && cbase0 == 0) && lean_tp_ok<TN>(a, cbase0)) {
                unsigned char* tw = reinterpret_cast<unsigned char*>(smem) + wave * LEAN_TP_BYTES<TN>;
                if (a.res != nullptr) finish_wave_tile_lean_tp<DT, TN, TM, true>(a, acc, cbase0, mbase, lane, tw);
                else finish_wave_tile_lean_tp<DT, TN, TM, false>(a, acc, cbase0, mbase, lane, tw);
                return;
            }
        }
        StoreEpilogue<DT, ODT>{a}(acc, mbase, cbase0, lane, wave, smem);
    }
};

template <int DT, int ODT, int BM, int BN, int WM, int WN, int STAGES, bool IS1X1, bool UTAP, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_v2_kernel(const ConvArgs a) {   // >= 2 waves per SIMD: <= 256 VGPR + AGPR
    conv_igemm_v2_body<DT, ODT, BM, BN, WM, WN, STAGES, IS1X1, UTAP, PIPE>(a, StoreEpilogue<DT, ODT>{a}, blockIdx.x);
}
template <int DT, int ODT, int BM, int BN, int WM, int WN, int STAGES, bool IS1X1, bool UTAP, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_v2_tp_kernel(const ConvArgs a) {   // the same tile with row-transposed stores (tiles 141-145)
    static_assert(STAGES * (BM + BN) * 64 >= 4 * LEAN_TP_BYTES<WN / 32>, "the operand ring holds the four waves' store tiles");
    conv_igemm_v2_body<DT, ODT, BM, BN, WM, WN, STAGES, IS1X1, UTAP, PIPE>(a, StoreEpilogueTP<DT, ODT>{a}, blockIdx.x);
}

// ---- detection head with the decode fused into the epilogue (head_decode.hpp): 128 pixels x (3 anchors x 32*TNA rows) ----
// operand ring depth of the head kernel.  NA = 3: the LDS footprint is set by the decode buffers anyway; NA = 1: two stages
// (28 KiB, the size of the decode buffers) leave room for five blocks per CU, which hide more latency than a third stage
template <int NA> constexpr int hd_stages() { return NA == 3 ? 3 : 2; }

template <int TNA, int NA>
struct DecodeEpilogue {
    const ConvArgs& a;
    const HeadDecodeArgs& h;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[NA * TNA][1], int mbase, int cbase0, int lane, int wave, uint16_t* smem) const {
        constexpr int HD_BUF = HdCfg<NA>::BUF, HD_WL = HdCfg<NA>::WL;
        if (a.debug & 4) return;   // tuning aid: convolution only
        __syncthreads();   // every wave is done with the operand ring: it becomes the record buffers
        uint64_t* bhi = reinterpret_cast<uint64_t*>(smem) + wave * HD_BUF;
        uint32_t* blo = reinterpret_cast<uint32_t*>(reinterpret_cast<uint64_t*>(smem) + 4 * HD_BUF) + wave * HD_BUF;
        u32x4* wl = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(smem) + 4 * HD_BUF * 12) + wave * HD_WL;
        head_decode_wave<TNA, NA>(a, h, acc, mbase + (lane & 31), lane, bhi, blo, wl, NA == 3 ? 0 : cbase0 / (32 * TNA));
    }
};

// NA = 3: one wave per SIMD (400 registers); NA = 1: three blocks per pixel tile, >= 3 waves per SIMD
template <int DT, int TNA, int NA>
__global__ __launch_bounds__(256, NA == 3 ? 1 : 3) void conv_head_decode_kernel(const ConvArgs a, const HeadDecodeArgs h) {
    conv_igemm_v2_body<DT, YMI_F32, 128, 32 * TNA * NA, 32, 32 * TNA * NA, hd_stages<NA>(), true, false, true>(a, DecodeEpilogue<TNA, NA>{a, h}, blockIdx.x);
}

// every pyramid level's head in ONE launch: the levels are independent and the coarse ones have few blocks (100 for a
// 20x20 map at batch 32), so back-to-back launches leave most of the chip idle -- block ranges select the level
struct HeadGroupArgs {
    ConvArgs a[YMI_MAX_LEVELS];
    HeadDecodeArgs h[YMI_MAX_LEVELS];
    int first_block[YMI_MAX_LEVELS + 1];
    int n;
};

template <int DT, int TNA, int NA>
__global__ __launch_bounds__(256, NA == 3 ? 1 : 3) void conv_head_decode_group_kernel(const HeadGroupArgs g) {
    // The level is a wave-uniform index into the argument block.  Indexing the by-value parameter copies the block to
    // scratch, and copying the selected level's structs with selects cost ~250 SALU per wave; reading the block through the
    // kernarg segment pointer (constant address space; the block is the kernel's only parameter, offset 0) makes every field
    // read one scalar load at a dynamic offset.
    int l = 0;
    static_for<1, YMI_MAX_LEVELS>([&](auto lt) {
        constexpr int lv = decltype(lt)::value;
        if (lv < g.n && (int)blockIdx.x >= g.first_block[lv] && (int)blockIdx.x < g.first_block[lv] + g.a[lv].nblk_m * g.a[lv].nblk_n) l = lv;
    });
    const HeadGroupArgs* gp = (const HeadGroupArgs*)(const __attribute__((address_space(4))) HeadGroupArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const ConvArgs& a = gp->a[l];
    const HeadDecodeArgs& h = gp->h[l];
    const int first = gp->first_block[l];
    conv_igemm_v2_body<DT, YMI_F32, 128, 32 * TNA * NA, 32, 32 * TNA * NA, hd_stages<NA>(), true, false, true>(a, DecodeEpilogue<TNA, NA>{a, h}, (int)blockIdx.x - first);
}

template <typename K>
int launch_v2_kernel(K kfn, const ConvArgs& a, size_t lds, dim3 grid, hipStream_t s) {
    if (lds < lds_floor_bytes()) lds = lds_floor_bytes();
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);
    return check_launch("conv_igemm_v2_kernel");
}

template <int DT, int ODT, int BM, int BN, int WM, int WN, int STAGES, bool PIPE = false, bool TP = false>
int launch_v2(const ConvArgs& a0, bool is1x1, hipStream_t s) {
    ConvArgs a = a0;
    a.nblk_m = cdiv(a.M, BM);
    a.nblk_n = cdiv(a.cout_pad, BN);
    if (a.chain_w != nullptr && !(BN == WN && BN == a.chain_k && BN <= 128)) {
        set_error("ymi_conv2d: this tile does not fit the chained 1x1 convolution (its cout width must equal %d)", a.chain_k);
        return YMI_EINVAL;
    }
    const bool utap = (a.cin % 32 == 0) && (a.kh * a.kw <= 32);
    const size_t lds = (size_t)STAGES * (BM + BN) * 64 + ((is1x1 || utap) ? 0 : (size_t)a.k_pad) + 16;
    dim3 grid(a.nblk_m * a.nblk_n);
    if constexpr (TP) {   // row-transposed stores: the unit-tap and pointwise forms only (what the big-map layers are)
        if (utap) return launch_v2_kernel(conv_igemm_v2_tp_kernel<DT, ODT, BM, BN, WM, WN, STAGES, false, true, PIPE>, a, lds, grid, s);
        if constexpr (!PIPE) {
            if (is1x1) return launch_v2_kernel(conv_igemm_v2_tp_kernel<DT, ODT, BM, BN, WM, WN, STAGES, true, false>, a, lds, grid, s);
        }
        set_error("ymi_conv2d: the row-transposed-store tiles need cin %% 32 == 0 (or a pointwise convolution)");
        return YMI_EINVAL;
    }
    if (utap) return launch_v2_kernel(conv_igemm_v2_kernel<DT, ODT, BM, BN, WM, WN, STAGES, false, true, PIPE>, a, lds, grid, s);
    if constexpr (PIPE) {
        set_error("ymi_conv2d: the software-pipelined tiles need cin %% 32 == 0");
        return YMI_EINVAL;
    } else {
        if (is1x1) return launch_v2_kernel(conv_igemm_v2_kernel<DT, ODT, BM, BN, WM, WN, STAGES, true, false>, a, lds, grid, s);
        return launch_v2_kernel(conv_igemm_v2_kernel<DT, ODT, BM, BN, WM, WN, STAGES, false, false>, a, lds, grid, s);
    }
}

template <int DT, int ODT, int BM, int BN, int WM, int WN>
int launch_cfg(const ConvArgs& a0, bool is1x1, hipStream_t s) {
    ConvArgs a = a0;
    a.nblk_m = cdiv(a.M, BM);
    a.nblk_n = cdiv(a.cout_pad, BN);
    dim3 grid(a.nblk_m * a.nblk_n), block(256);
    if (is1x1)
        hipLaunchKernelGGL((conv_igemm_kernel<DT, ODT, BM, BN, WM, WN, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<DT, ODT, BM, BN, WM, WN, false>), grid, block, 0, s, a);
    return check_launch("conv_igemm_kernel");
}

// tile ids: 1 = 128x128, 2 = 256x64, 3 = 256x32, 4 = 64x128, 5 = 128x64, 6 = 64x64... keep small
// The tile instantiations are spread over several translation units (conv_inst_*.hip: one per tile group and dtype pair)
// so that the library builds in parallel; conv_igemm.hip only dispatches.
//   group 0: register-staged kernel (1..5) + 3/4-stage LDS-DMA tiles (11..15)      group 1: 2-stage tiles (21..27)
//   group 2: software-pipelined 61..70                                              group 3: software-pipelined 71..81
//   group 4: 141..145 = tiles 12 / 21 / 66 / 61 / 71 with row-transposed stores (opt-in, not in the pinned table yet)
template <int G, int DT, int ODT>
int launch_tile_group(const ConvArgs& a, bool is1x1, int tile, hipStream_t s) {
    if constexpr (G == 0) {
        switch (tile) {
            case 11: return launch_v2<DT, ODT, 128, 128, 64, 64, 4>(a, is1x1, s);
            case 12: return launch_v2<DT, ODT, 256, 64, 64, 64, 3>(a, is1x1, s);
            case 13: return launch_v2<DT, ODT, 256, 32, 64, 32, 3>(a, is1x1, s);
            case 14: return launch_v2<DT, ODT, 64, 128, 32, 64, 4>(a, is1x1, s);
            case 15: return launch_v2<DT, ODT, 128, 64, 64, 32, 4>(a, is1x1, s);
            case 1: return launch_cfg<DT, ODT, 128, 128, 64, 64>(a, is1x1, s);
            case 2: return launch_cfg<DT, ODT, 256, 64, 64, 64>(a, is1x1, s);
            case 3: return launch_cfg<DT, ODT, 256, 32, 64, 32>(a, is1x1, s);
            case 4: return launch_cfg<DT, ODT, 64, 128, 32, 64>(a, is1x1, s);
            case 5: return launch_cfg<DT, ODT, 128, 64, 64, 32>(a, is1x1, s);
            default: break;
        }
    }
    if constexpr (G == 1) {
        switch (tile) {
            case 21: return launch_v2<DT, ODT, 128, 128, 64, 64, 2>(a, is1x1, s);
            case 22: return launch_v2<DT, ODT, 256, 64, 64, 64, 2>(a, is1x1, s);
            case 23: return launch_v2<DT, ODT, 256, 32, 64, 32, 2>(a, is1x1, s);
            case 24: return launch_v2<DT, ODT, 64, 128, 32, 64, 2>(a, is1x1, s);
            case 25: return launch_v2<DT, ODT, 128, 64, 64, 32, 2>(a, is1x1, s);
            case 26: return launch_v2<DT, ODT, 128, 32, 32, 32, 2>(a, is1x1, s);
            case 27: return launch_v2<DT, ODT, 64, 64, 32, 32, 2>(a, is1x1, s);
            default: break;
        }
    }
    if constexpr (G == 2) {
        switch (tile) {
            case 61: return launch_v2<DT, ODT, 128, 128, 64, 64, 4, true>(a, is1x1, s);
            case 62: return launch_v2<DT, ODT, 256, 64, 64, 64, 3, true>(a, is1x1, s);
            case 63: return launch_v2<DT, ODT, 256, 32, 64, 32, 3, true>(a, is1x1, s);
            case 64: return launch_v2<DT, ODT, 64, 128, 32, 64, 4, true>(a, is1x1, s);
            case 65: return launch_v2<DT, ODT, 128, 64, 64, 32, 4, true>(a, is1x1, s);
            case 66: return launch_v2<DT, ODT, 256, 128, 128, 64, 3, true>(a, is1x1, s);
            case 68: return launch_v2<DT, ODT, 256, 128, 128, 64, 2, true>(a, is1x1, s);
            case 69: return launch_v2<DT, ODT, 128, 128, 64, 64, 3, true>(a, is1x1, s);
            case 70: return launch_v2<DT, ODT, 128, 64, 64, 32, 3, true>(a, is1x1, s);
            default: break;
        }
    }
    if constexpr (G == 3) {
        switch (tile) {
            case 78: return launch_v2<DT, ODT, 128, 128, 32, 128, 2, true>(a, is1x1, s);
            case 79: return launch_v2<DT, ODT, 128, 128, 32, 128, 3, true>(a, is1x1, s);
            case 80: return launch_v2<DT, ODT, 128, 64, 32, 64, 2, true>(a, is1x1, s);
            case 81: return launch_v2<DT, ODT, 128, 64, 32, 64, 3, true>(a, is1x1, s);
            case 71: return launch_v2<DT, ODT, 128, 128, 64, 64, 2, true>(a, is1x1, s);
            case 72: return launch_v2<DT, ODT, 256, 64, 64, 64, 2, true>(a, is1x1, s);
            case 73: return launch_v2<DT, ODT, 256, 32, 64, 32, 2, true>(a, is1x1, s);
            case 74: return launch_v2<DT, ODT, 64, 128, 32, 64, 2, true>(a, is1x1, s);
            case 75: return launch_v2<DT, ODT, 128, 64, 64, 32, 2, true>(a, is1x1, s);
            case 76: return launch_v2<DT, ODT, 128, 32, 32, 32, 2, true>(a, is1x1, s);
            case 77: return launch_v2<DT, ODT, 64, 64, 32, 32, 2, true>(a, is1x1, s);
            default: break;
        }
    }
    if constexpr (G == 4) {   // tiles 12 / 21 / 66 / 61 / 71 with row-transposed stores (StoreEpilogueTP); 16-bit outputs
        if constexpr (ODT == DT) {
            switch (tile) {
                case 141: return launch_v2<DT, ODT, 256, 64, 64, 64, 3, false, true>(a, is1x1, s);
                case 142: return launch_v2<DT, ODT, 128, 128, 64, 64, 2, false, true>(a, is1x1, s);
                case 143: return launch_v2<DT, ODT, 256, 128, 128, 64, 3, true, true>(a, is1x1, s);
                case 144: return launch_v2<DT, ODT, 128, 128, 64, 64, 4, true, true>(a, is1x1, s);
                case 145: return launch_v2<DT, ODT, 128, 128, 64, 64, 2, true, true>(a, is1x1, s);
                default: break;
            }
        } else {
            set_error("ymi_conv2d: the row-transposed-store tiles write the compute dtype");
            return YMI_EINVAL;
        }
    }
    set_error("ymi_conv2d: unknown tile id %d", tile);
    return YMI_EINVAL;
}

inline int tile_group_of(int tile) { return (tile >= 141 && tile <= 149) ? 4 : (tile <= 5 || (tile >= 11 && tile <= 15)) ? 0 : (tile >= 21 && tile <= 27) ? 1 : (tile >= 61 && tile <= 70) ? 2 : 3; }

template <int NA>
inline size_t head_decode_lds(int tna) {
    size_t lds = (size_t)hd_stages<NA>() * (128 + 32 * tna * NA) * 64 + 16;
    const size_t need = (size_t)HdCfg<NA>::LDS_BYTES + 16;   // the ring doubles as the per-wave record buffers and worklists
    return lds < need ? need : lds;
}

// the launchers take the anchor split from ConvArgs::nblk_n (1: a wave holds all three anchors, 3: one anchor per block)
template <int DT, int TNA>
int launch_head_decode(const ConvArgs& a, const HeadDecodeArgs& h, hipStream_t s) {
    if (a.nblk_n == 3) {
        const size_t lds = head_decode_lds<1>(TNA);
        auto kfn = conv_head_decode_kernel<DT, TNA, 1>;
        if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
        hipLaunchKernelGGL(kfn, dim3(a.nblk_m * 3), dim3(256), lds, s, a, h);
        return check_launch("conv_head_decode_kernel");
    }
    const size_t lds = head_decode_lds<3>(TNA);
    auto kfn = conv_head_decode_kernel<DT, TNA, 3>;
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    hipLaunchKernelGGL(kfn, dim3(a.nblk_m), dim3(256), lds, s, a, h);
    return check_launch("conv_head_decode_kernel");
}

template <int DT, int TNA>
int launch_head_group(const HeadGroupArgs& g, hipStream_t s) {
    if (g.a[0].nblk_n == 3) {
        const size_t lds = head_decode_lds<1>(TNA);
        auto kfn = conv_head_decode_group_kernel<DT, TNA, 1>;
        if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
        hipLaunchKernelGGL(kfn, dim3(g.first_block[g.n]), dim3(256), lds, s, g);
        return check_launch("conv_head_decode_group_kernel");
    }
    const size_t lds = head_decode_lds<3>(TNA);
    auto kfn = conv_head_decode_group_kernel<DT, TNA, 3>;
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    hipLaunchKernelGGL(kfn, dim3(g.first_block[g.n]), dim3(256), lds, s, g);
    return check_launch("conv_head_decode_group_kernel");
}

}  // namespace ymi
