// Implicit-GEMM convolution for gfx950 (MI355X): NHWC fp16/bf16 activations, MFMA 32x32x16,
// fused bias + SiLU (+ residual) epilogue, reads/writes channel slices of concat buffers.
//
// GEMM view (computed "swapped" so that each lane ends up owning consecutive output channels of
// ONE pixel, which makes the NHWC store 8/16 bytes wide):
//     D[cout][pixel] = sum_k  W[cout][k] * X[pixel][k],   k = (ky*kw + kx)*cin + c
//   MFMA A operand = weight fragment (rows = cout), B operand = activation fragment (cols = pixel).
//   v_mfma_f32_32x32x16: lane l holds A[row = l&31][k = 8*(l>>5) .. +7], B[k = 8*(l>>5)..+7][col = l&31],
//   D reg r of lane l = D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
//
// Tiling: block = 256 threads = 4 waves, block tile BM pixels x BN couts, BK = 32 per step.
// Activation tile is gathered straight from the NHWC input with 16-byte loads (8 channels of one
// tap of one pixel), zero-filled outside the image; an int32 table (ktab) gives, per 8-channel
// k-chunk, the element offset and (dy,dx) of its tap, so 1x1, 3x3 s1/s2 and the 6x6 s2 stem share
// one kernel.  Tiles are double-buffered in LDS (register-staged: global loads for step t+1 are
// issued before the MFMAs of step t; the LDS write lands after them).  LDS rows are padded to
// 80 bytes: a 16-lane ds_read_b128 group then touches 16 distinct 16-byte slots (conflict free).
//
// Replaces: yolort/v5/models/common.py:69-70 (Conv.forward: conv2d -> BatchNorm2d -> SiLU, BN folded),
//           common.py:115-116 (Bottleneck residual), yolort/models/box_head.py:36,74 (head conv).
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
                        o[0] = (uint32_t)to16<DT>(v[0]) | ((uint32_t)to16<DT>(v[1]) << 16);
                        o[1] = (uint32_t)to16<DT>(v[2]) | ((uint32_t)to16<DT>(v[3]) << 16);
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
    const int nsteps = a.k_pad / BK;
    f32x4 bias_regs[TN][4];   // issued first: the latency hides behind the geometry math below
    load_bias<TN>(a, n0 + wave_n, lane >> 5, bias_regs);

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
    init_acc<TN, TM>(acc, bias_regs);   // accumulate on top of the bias

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

template <int DT, int ODT, int BM, int BN, int WM, int WN, int STAGES, bool IS1X1, bool UTAP, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_v2_kernel(const ConvArgs a) {   // >= 2 waves per SIMD: <= 256 VGPR + AGPR
    conv_igemm_v2_body<DT, ODT, BM, BN, WM, WN, STAGES, IS1X1, UTAP, PIPE>(a, StoreEpilogue<DT, ODT>{a}, blockIdx.x);
}

// ---- detection head with the decode fused into the epilogue (head_decode.hpp): 128 pixels x (3 anchors x 32*TNA rows) ----
constexpr int HD_STAGES = 3;   // operand ring depth of the head kernel (its LDS footprint is set by the decode buffers anyway)

template <int TNA>
struct DecodeEpilogue {
    const ConvArgs& a;
    const HeadDecodeArgs& h;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[3 * TNA][1], int mbase, int, int lane, int wave, uint16_t* smem) const {
        __syncthreads();   // every wave is done with the operand ring: it becomes the record buffers
        uint64_t* bhi = reinterpret_cast<uint64_t*>(smem) + wave * HD_BUF;
        uint32_t* blo = reinterpret_cast<uint32_t*>(reinterpret_cast<uint64_t*>(smem) + 4 * HD_BUF) + wave * HD_BUF;
        u32x4* wl = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(smem) + 4 * HD_BUF * 12) + wave * HD_WL;
        head_decode_wave<TNA>(a, h, acc, mbase + (lane & 31), lane, bhi, blo, wl);
    }
};

template <int DT, int TNA>
__global__ __launch_bounds__(256) void conv_head_decode_kernel(const ConvArgs a, const HeadDecodeArgs h) {
    conv_igemm_v2_body<DT, YMI_F32, 128, 96 * TNA, 32, 96 * TNA, HD_STAGES, false, true, true>(a, DecodeEpilogue<TNA>{a, h}, blockIdx.x);
}

// every pyramid level's head in ONE launch: the levels are independent and the coarse ones have few blocks (100 for a
// 20x20 map at batch 32), so back-to-back launches leave most of the chip idle -- block ranges select the level
struct HeadGroupArgs {
    ConvArgs a[YMI_MAX_LEVELS];
    HeadDecodeArgs h[YMI_MAX_LEVELS];
    int first_block[YMI_MAX_LEVELS + 1];
    int n;
};

template <int DT, int TNA>
__global__ __launch_bounds__(256) void conv_head_decode_group_kernel(const HeadGroupArgs g) {
    // constant indices only: a runtime index into the by-value argument block would copy it to scratch
    ConvArgs a = g.a[0];
    HeadDecodeArgs h = g.h[0];
    int first = g.first_block[0];
    static_for<1, YMI_MAX_LEVELS>([&](auto lt) {
        constexpr int l = decltype(lt)::value;
        if (l < g.n && (int)blockIdx.x >= g.first_block[l] && (int)blockIdx.x < g.first_block[l] + g.a[l].nblk_m) {   // wave-uniform
            a = g.a[l];
            h = g.h[l];
            first = g.first_block[l];
        }
    });
    conv_igemm_v2_body<DT, YMI_F32, 128, 96 * TNA, 32, 96 * TNA, HD_STAGES, false, true, true>(a, DecodeEpilogue<TNA>{a, h}, (int)blockIdx.x - first);
}

template <typename K>
static int launch_v2_kernel(K kfn, const ConvArgs& a, size_t lds, dim3 grid, hipStream_t s) {
    if (lds > 64 * 1024) YMI_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);
    return check_launch("conv_igemm_v2_kernel");
}

template <int DT, int ODT, int BM, int BN, int WM, int WN, int STAGES, bool PIPE = false>
static int launch_v2(const ConvArgs& a0, bool is1x1, hipStream_t s) {
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
static int launch_cfg(const ConvArgs& a0, bool is1x1, hipStream_t s) {
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
template <int DT, int ODT>
static int launch_dtype(const ConvArgs& a0, bool is1x1, int tile, hipStream_t s) {
    ConvArgs a = a0;
    a.debug = 0;
    if (tile >= 0x100) { a.debug = tile >> 8; tile &= 0xff; }
    const bool force_v1 = tile < 0;
    if (force_v1) tile = -tile == 100 ? 0 : -tile;   // negative tile ids force the register-staged kernel (-100 = auto)
    if (a.chain_w != nullptr && (force_v1 || (tile >= 1 && tile <= 5 && a.zeros == nullptr) || tile == 41)) {
        set_error("ymi_conv2d: the chained 1x1 convolution needs the pipelined implicit-GEMM kernel");
        return YMI_EINVAL;
    }
    if (tile == 0 && a.chain_w != nullptr) tile = a.chain_k == 32 ? 3 : (a.chain_k == 64 ? 2 : 78);   // cout width == chain K, four waves along the pixels
    if (tile == 0) {
        const int cp = a.cout_pad;
        if (cp <= 32) tile = 3;
        else if (cp <= 64) tile = 2;
        else {
            // enough 128x128 tiles to fill 256 CUs a few times over? else halve the pixel tile
            const long blocks = (long)cdiv(a.M, 128) * cdiv(cp, 128);
            tile = (blocks >= 512) ? 1 : 4;
            if (cp % 128 != 0 && cp % 64 == 0 && cp < 128) tile = 2;
        }
    }
    if (tile < 10 && a.zeros != nullptr && !force_v1) tile += 10;   // pipelined kernel whenever a zero page is supplied
    if (tile >= 10 && a.zeros == nullptr) {
        set_error("ymi_conv2d: tile %d (pipelined kernel) needs desc.zeros", tile);
        return YMI_EINVAL;
    }
    if (tile >= 31 && tile <= 39) return conv3x3_halo_launch(a, DT, ODT, tile - 30, s);   // LDS-halo 3x3 s1 kernel
    if (tile == 41) return conv_stem_launch(a, DT, ODT, s);
    switch (tile) {
        case 11: return launch_v2<DT, ODT, 128, 128, 64, 64, 4>(a, is1x1, s);
        case 12: return launch_v2<DT, ODT, 256, 64, 64, 64, 3>(a, is1x1, s);
        case 13: return launch_v2<DT, ODT, 256, 32, 64, 32, 3>(a, is1x1, s);
        case 14: return launch_v2<DT, ODT, 64, 128, 32, 64, 4>(a, is1x1, s);
        case 15: return launch_v2<DT, ODT, 128, 64, 64, 32, 4>(a, is1x1, s);
        // 2-stage rings: half the LDS, twice the resident blocks -- for short K (1x1 convs, stem) where the
        // ring never fills and occupancy hides the operand latency instead
        case 21: return launch_v2<DT, ODT, 128, 128, 64, 64, 2>(a, is1x1, s);
        case 22: return launch_v2<DT, ODT, 256, 64, 64, 64, 2>(a, is1x1, s);
        case 23: return launch_v2<DT, ODT, 256, 32, 64, 32, 2>(a, is1x1, s);
        case 24: return launch_v2<DT, ODT, 64, 128, 32, 64, 2>(a, is1x1, s);
        case 25: return launch_v2<DT, ODT, 128, 64, 64, 32, 2>(a, is1x1, s);
        case 26: return launch_v2<DT, ODT, 128, 32, 32, 32, 2>(a, is1x1, s);   // 20 KB LDS: 8 blocks / CU
        case 27: return launch_v2<DT, ODT, 64, 64, 32, 32, 2>(a, is1x1, s);    // 16 KB LDS
        // software-pipelined main loop (61..65 = 11..15, 71..77 = 21..27)
        case 61: return launch_v2<DT, ODT, 128, 128, 64, 64, 4, true>(a, is1x1, s);
        case 62: return launch_v2<DT, ODT, 256, 64, 64, 64, 3, true>(a, is1x1, s);
        case 63: return launch_v2<DT, ODT, 256, 32, 64, 32, 3, true>(a, is1x1, s);
        case 64: return launch_v2<DT, ODT, 64, 128, 32, 64, 4, true>(a, is1x1, s);
        case 65: return launch_v2<DT, ODT, 128, 64, 64, 32, 4, true>(a, is1x1, s);
        case 66: return launch_v2<DT, ODT, 256, 128, 128, 64, 3, true>(a, is1x1, s);   // 256-pixel tiles: fewer DMA pieces per MFMA
        case 68: return launch_v2<DT, ODT, 256, 128, 128, 64, 2, true>(a, is1x1, s);
        case 69: return launch_v2<DT, ODT, 128, 128, 64, 64, 3, true>(a, is1x1, s);    // 48 KB LDS: 3 blocks / CU
        case 70: return launch_v2<DT, ODT, 128, 64, 64, 32, 3, true>(a, is1x1, s);
        case 78: return launch_v2<DT, ODT, 128, 128, 32, 128, 2, true>(a, is1x1, s);   // four waves along the pixels, each all 128 couts (chained 1x1, K = 128)
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
        case 1: return launch_cfg<DT, ODT, 128, 128, 64, 64>(a, is1x1, s);
        case 2: return launch_cfg<DT, ODT, 256, 64, 64, 64>(a, is1x1, s);
        case 3: return launch_cfg<DT, ODT, 256, 32, 64, 32>(a, is1x1, s);
        case 4: return launch_cfg<DT, ODT, 64, 128, 32, 64>(a, is1x1, s);
        case 5: return launch_cfg<DT, ODT, 128, 64, 64, 32>(a, is1x1, s);
        default: set_error("ymi_conv2d: unknown tile id %d", tile); return YMI_EINVAL;
    }
}

// descriptor -> kernel arguments (shared by ymi_conv2d and ymi_conv_head_decode); validation of the zero page included
static int fill_conv_args(const ymi_conv_desc* d, ConvArgs& a) {
    a.x = (const uint16_t*)d->x; a.w = (const uint16_t*)d->w; a.bias = d->bias; a.ktab = (const int2*)d->ktab;
    a.y = d->y; a.res = (const uint16_t*)d->res;
    a.n = d->n; a.h = d->h; a.w_in = d->w_in; a.cin = d->cin; a.x_cs = d->x_cstride;
    a.ho = d->ho; a.wo = d->wo; a.cout = d->cout; a.cout_pad = d->cout_pad; a.y_cs = d->y_cstride; a.res_cs = d->res_cstride;
    a.sh = d->sh; a.sw = d->sw; a.ph = d->ph; a.pw = d->pw; a.k_pad = d->k_pad; a.act = d->act;
    a.M = d->n * d->ho * d->wo; a.nblk_m = 0; a.nblk_n = 0;
    a.y2 = d->y2; a.y2_cs = d->y2_cstride; a.split = d->cout_split; a.zeros = (const uint16_t*)d->zeros;
    a.up2 = d->y2_mode == 1 ? 1 : 0;
    a.chain_w = (const uint16_t*)d->chain_w; a.chain_bias = d->chain_bias; a.chain_y = d->chain_y;
    a.chain_cout = d->chain_cout; a.chain_y_cs = d->chain_y_cstride; a.chain_k = d->cout_split > 0 ? d->cout_split : d->cout;
    a.chain_x2 = (const uint16_t*)d->chain_x2; a.chain_x2_cs = d->chain_x2_cstride; a.chain_k2 = d->chain_x2 != nullptr ? d->chain_k2 : 0;
    a.kh = d->kh; a.kw = d->kw; a.x_zero_off = 0;
    auto magic = [](int dv) { const uint64_t v = (((uint64_t)1 << 32) / (uint64_t)dv) + 1u; return (unsigned)(v > 0xffffffffull ? 0xffffffffull : v); };
    a.magic_hw = magic(d->ho * d->wo);
    a.magic_w = magic(d->wo);
    if (d->zeros != nullptr) {
        const int64_t dz = ((const char*)d->zeros - (const char*)d->x) / 2;
        YMI_REQUIRE(dz > -((int64_t)1 << 31) && dz < ((int64_t)1 << 31) && ((const char*)d->zeros - (const char*)d->x) % 16 == 0,
                    "ymi_conv2d: desc.zeros must lie within +-4 GiB of x and be 16-byte aligned relative to it (use the tail of x's buffer)");
        a.x_zero_off = (int)dz;
    }
    return YMI_OK;
}

int conv2d_launch(const ymi_conv_desc* d, hipStream_t s) {
    YMI_REQUIRE(d != nullptr, "ymi_conv2d: null descriptor");
    YMI_REQUIRE(d->x && d->w && d->bias && d->y, "ymi_conv2d: null buffer");
    YMI_REQUIRE(d->cin % 8 == 0 && d->x_cstride % 8 == 0, "ymi_conv2d: cin (%d) and x_cstride (%d) must be multiples of 8", d->cin, d->x_cstride);
    YMI_REQUIRE(d->cout_pad % 32 == 0 && d->k_pad % 32 == 0, "ymi_conv2d: cout_pad (%d) / k_pad (%d) must be multiples of 32", d->cout_pad, d->k_pad);
    YMI_REQUIRE(d->k_pad >= d->kh * d->kw * d->cin, "ymi_conv2d: k_pad %d < K %d", d->k_pad, d->kh * d->kw * d->cin);
    YMI_REQUIRE(d->y_cstride % 4 == 0 && (d->res == nullptr || d->res_cstride % 4 == 0), "ymi_conv2d: y/res cstride must be multiples of 4");
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16, "ymi_conv2d: dtype must be F16 or BF16");
    YMI_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == YMI_F32, "ymi_conv2d: out_dtype must equal dtype or be F32");
    YMI_REQUIRE(d->ho == (d->h + 2 * d->ph - d->kh) / d->sh + 1 && d->wo == (d->w_in + 2 * d->pw - d->kw) / d->sw + 1,
                "ymi_conv2d: output size %dx%d inconsistent with input %dx%d k%dx%d s%dx%d p%dx%d", d->ho, d->wo, d->h, d->w_in, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
    const bool is1x1 = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0;
    YMI_REQUIRE(is1x1 || d->ktab != nullptr, "ymi_conv2d: ktab required for non-1x1 convolutions");
    if ((int64_t)d->n * d->ho * d->wo >= (int64_t)1 << 31) {
        set_error("ymi_conv2d: too many output pixels");
        return YMI_EINVAL;
    }
    ConvArgs a;
    { const int rc_args = fill_conv_args(d, a); if (rc_args != YMI_OK) return rc_args; }
    YMI_REQUIRE(a.split == 0 || (d->y2 != nullptr && a.split % 8 == 0 && a.split < d->cout && d->res == nullptr && d->out_dtype == d->dtype && d->y2_cstride % 8 == 0),
                "ymi_conv2d: invalid second-output configuration");
    YMI_REQUIRE(a.split == 0 || a.zeros != nullptr, "ymi_conv2d: the second output needs the pipelined kernel (desc.zeros)");
    YMI_REQUIRE(d->y2_mode == 0 || d->y2_mode == 1, "ymi_conv2d: unknown y2_mode %d", d->y2_mode);
    if (d->chain_w != nullptr) {
        const int k1 = d->cout_split > 0 ? d->cout_split : d->cout;
        YMI_REQUIRE(d->chain_bias && d->chain_y && (k1 == 32 || k1 == 64 || k1 == 128) && d->chain_cout % 32 == 0 && d->chain_cout >= 32 && d->chain_cout <= 128 &&
                        d->chain_y_cstride % 8 == 0 && d->act == YMI_ACT_SILU && d->out_dtype == d->dtype && d->y2_mode == 0 && a.zeros != nullptr,
                    "ymi_conv2d: chained 1x1 needs chain_bias / chain_y, K1 in {32, 64, 128}, chain_cout %% 32 == 0 (<= 128), SiLU, a 16-bit output, desc.zeros");
        YMI_REQUIRE(d->chain_x2 == nullptr || (d->chain_k2 % 16 == 0 && d->chain_k2 >= 16 && d->chain_k2 <= 128 && d->chain_x2_cstride % 8 == 0 && d->cout_split == 0),
                    "ymi_conv2d: chained conv second source: chain_k2 %% 16 == 0 (16..128), chain_x2_cstride %% 8 == 0, no channel split");
        YMI_REQUIRE(d->res == nullptr || d->chain_x2 != nullptr || true, "ymi_conv2d: internal");
    }
    YMI_REQUIRE(d->y2_mode == 0 || (d->y2 != nullptr && a.split == 0 && d->cout % 32 == 0 && d->out_dtype == d->dtype && d->y2_cstride % 8 == 0 && a.zeros != nullptr),
                "ymi_conv2d: the upsampled second output needs y2, cout_split == 0, cout %% 32 == 0, a 16-bit output, y2_cstride %% 8 == 0 and desc.zeros");
    if (d->y2_mode == 1) {
        YMI_REQUIRE(d->tile >= 0, "ymi_conv2d: the upsampled second output is not available in the register-staged kernel");
        YMI_REQUIRE((int64_t)d->n * 4 * d->ho * d->wo < ((int64_t)1 << 31), "ymi_conv2d: upsampled view too large");
    }
    if (a.zeros != nullptr) {
        // 32-bit element offsets inside the pipelined kernel
        YMI_REQUIRE((int64_t)d->n * d->h * d->w_in * d->x_cstride < ((int64_t)1 << 31) && (int64_t)d->cout_pad * d->k_pad < ((int64_t)1 << 31),
                    "ymi_conv2d: tensor too large for the pipelined kernel's 32-bit offsets");
        YMI_REQUIRE(d->y_cstride % 8 == 0 || d->out_dtype == YMI_F32, "ymi_conv2d: y_cstride must be a multiple of 8");
    }
    if (a.M == 0) return YMI_OK;
    if (d->dtype == YMI_F16) {
        if (d->out_dtype == YMI_F32) return launch_dtype<YMI_F16, YMI_F32>(a, is1x1, d->tile, s);
        return launch_dtype<YMI_F16, YMI_F16>(a, is1x1, d->tile, s);
    } else {
        if (d->out_dtype == YMI_F32) return launch_dtype<YMI_BF16, YMI_F32>(a, is1x1, d->tile, s);
        return launch_dtype<YMI_BF16, YMI_BF16>(a, is1x1, d->tile, s);
    }
}


// validation + argument construction of one level's fused head (shared by the single and the grouped launch)
static int head_decode_prepare(const ymi_conv_desc* d, const ymi_post_desc* post, int level, ConvArgs& a, HeadDecodeArgs& h, int& tna) {
    YMI_REQUIRE(d != nullptr && post != nullptr, "ymi_conv_head_decode: null descriptor");
    YMI_REQUIRE(level >= 0 && level < post->num_levels && post->num_levels <= YMI_MAX_LEVELS, "ymi_conv_head_decode: level %d out of range", level);
    YMI_REQUIRE(d->x && d->w && d->bias && d->zeros, "ymi_conv_head_decode: null buffer (x, w, bias and the zero page are required)");
    const int K = post->num_classes + 5;
    const int ra = (K + 31) / 32 * 32;
    YMI_REQUIRE(ra <= 128, "ymi_conv_head_decode: %d outputs per anchor exceed 128 (use ymi_conv2d + ymi_postprocess)", K);
    YMI_REQUIRE(d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && d->cin % 32 == 0 && d->k_pad == d->cin,
                "ymi_conv_head_decode: a 1x1 stride-1 convolution with cin %% 32 == 0 and k_pad == cin is required");
    YMI_REQUIRE(d->cout == 3 * ra && d->cout_pad == 3 * ra, "ymi_conv_head_decode: cout / cout_pad must be 3 x %d (anchor-padded packing)", ra);
    YMI_REQUIRE(d->res == nullptr && d->cout_split == 0 && d->act == YMI_ACT_NONE && d->chain_w == nullptr && d->y2_mode == 0,
                "ymi_conv_head_decode: no residual / second output / chained conv / activation");
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16, "ymi_conv_head_decode: dtype must be F16 or BF16");
    YMI_REQUIRE(d->x_cstride % 8 == 0, "ymi_conv_head_decode: x_cstride must be a multiple of 8");
    YMI_REQUIRE(d->n == post->n && d->ho == post->lh[level] && d->wo == post->lw[level] && d->h == d->ho && d->w_in == d->wo,
                "ymi_conv_head_decode: conv geometry %dx%dx%d does not match level %d of the post-process (%dx%dx%d)", d->n, d->ho, d->wo, level, post->n,
                post->lh[level], post->lw[level]);
    YMI_REQUIRE((int64_t)d->n * d->h * d->w_in * d->x_cstride < ((int64_t)1 << 31) && (int64_t)d->n * d->ho * d->wo < ((int64_t)1 << 31),
                "ymi_conv_head_decode: tensor too large for 32-bit offsets");
    YMI_REQUIRE(post->status && post->ws && post->n >= 1 && post->cand_cap >= 1, "ymi_conv_head_decode: incomplete post-process descriptor");
    const PostLayout L = post_layout(post);
    YMI_REQUIRE(L.label_bits + L.anchor_bits <= 32, "ymi_conv_head_decode: candidate index exceeds 32 bits");
    const Workspace w = carve(post->ws, post->n, L.total_anchors, post->cand_cap);
    YMI_REQUIRE(post->ws_bytes >= w.total, "ymi_conv_head_decode: workspace too small");
    { const int rc_args = fill_conv_args(d, a); if (rc_args != YMI_OK) return rc_args; }
    a.nblk_m = cdiv(a.M, 128);
    a.nblk_n = 1;
    h.stride = post->stride[level];
    for (int k = 0; k < 6; ++k) h.anc[k] = post->anchors[level][k];
    h.K = K;
    h.level_off = 0;
    for (int l = 0; l < level; ++l) h.level_off += 3 * post->lh[l] * post->lw[l];
    h.sink = make_sink(post, w, L);
    tna = ra / 32;
    return YMI_OK;
}

static size_t head_decode_lds(int tna) {
    size_t lds = (size_t)HD_STAGES * (128 + 96 * tna) * 64 + 16;
    const size_t need = (size_t)HD_LDS_BYTES + 16;   // the ring doubles as the per-wave record buffers and worklists
    return lds < need ? need : lds;
}

template <int DT, int TNA>
static int launch_head_decode(const ConvArgs& a, const HeadDecodeArgs& h, hipStream_t s) {
    const size_t lds = head_decode_lds(TNA);
    auto kfn = conv_head_decode_kernel<DT, TNA>;
    if (lds > 64 * 1024) YMI_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kfn, dim3(a.nblk_m), dim3(256), lds, s, a, h);
    return check_launch("conv_head_decode_kernel");
}

template <int DT, int TNA>
static int launch_head_group(const HeadGroupArgs& g, hipStream_t s) {
    const size_t lds = head_decode_lds(TNA);
    auto kfn = conv_head_decode_group_kernel<DT, TNA>;
    if (lds > 64 * 1024) YMI_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kfn, dim3(g.first_block[g.n]), dim3(256), lds, s, g);
    return check_launch("conv_head_decode_group_kernel");
}

#define YMI_HD_DISPATCH(FN, DTYPE, TNA_, ...)                                                              \
    switch (TNA_) {                                                                                        \
        case 1: return DTYPE == YMI_F16 ? FN<YMI_F16, 1>(__VA_ARGS__) : FN<YMI_BF16, 1>(__VA_ARGS__);     \
        case 2: return DTYPE == YMI_F16 ? FN<YMI_F16, 2>(__VA_ARGS__) : FN<YMI_BF16, 2>(__VA_ARGS__);     \
        case 3: return DTYPE == YMI_F16 ? FN<YMI_F16, 3>(__VA_ARGS__) : FN<YMI_BF16, 3>(__VA_ARGS__);     \
        case 4: return DTYPE == YMI_F16 ? FN<YMI_F16, 4>(__VA_ARGS__) : FN<YMI_BF16, 4>(__VA_ARGS__);     \
    }

// Head conv of pyramid level `level` with decode + threshold fused into the epilogue.  The descriptor's weights hold
// every anchor's K rows padded to RA = round_up(K, 32) rows (cout = cout_pad = 3*RA); y is not written.
int conv_head_decode_launch(const ymi_conv_desc* d, const ymi_post_desc* post, int level, hipStream_t s) {
    ConvArgs a;
    HeadDecodeArgs h;
    int tna = 0;
    const int rc = head_decode_prepare(d, post, level, a, h, tna);
    if (rc != YMI_OK) return rc;
    if (a.M == 0) return YMI_OK;
    YMI_HD_DISPATCH(launch_head_decode, d->dtype, tna, a, h, s)
    set_error("ymi_conv_head_decode: unsupported anchor padding");
    return YMI_EINVAL;
}

// all levels in one launch (descs[l] belongs to level l of `post`)
int conv_head_decode_group_launch(const ymi_conv_desc* descs, int n_levels, const ymi_post_desc* post, hipStream_t s) {
    YMI_REQUIRE(descs != nullptr && post != nullptr && n_levels >= 1 && n_levels <= YMI_MAX_LEVELS && n_levels == post->num_levels,
                "ymi_conv_head_decode_group: one descriptor per pyramid level of the post-process is required");
    HeadGroupArgs g;
    memset(&g, 0, sizeof(g));
    g.n = n_levels;
    int tna0 = 0, blocks = 0;
    for (int l = 0; l < n_levels; ++l) {
        int tna = 0;
        const int rc = head_decode_prepare(&descs[l], post, l, g.a[l], g.h[l], tna);
        if (rc != YMI_OK) return rc;
        YMI_REQUIRE(descs[l].dtype == descs[0].dtype && (l == 0 || tna == tna0), "ymi_conv_head_decode_group: levels must share dtype and class count");
        tna0 = tna;
    }
    // coarse levels first: they have the longest K loops and the fewest blocks, so they should not form the tail
    for (int l = n_levels - 1; l >= 0; --l) {
        g.first_block[l] = blocks;
        blocks += g.a[l].nblk_m;
    }
    g.first_block[n_levels] = blocks;
    if (blocks == 0) return YMI_OK;
    YMI_HD_DISPATCH(launch_head_group, descs[0].dtype, tna0, g, s)
    set_error("ymi_conv_head_decode_group: unsupported anchor padding");
    return YMI_EINVAL;
}
#undef YMI_HD_DISPATCH

}  // namespace ymi

#ifdef YMI_STAMPS
extern "C" int ymi_debug_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ymi::ymi_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ymi_conv2d(const ymi_conv_desc* d, void* stream) { return ymi::conv2d_launch(d, (hipStream_t)stream); }
extern "C" int ymi_conv_stem_planar(const ymi_conv_desc* d, const void* const* imgs, int n_imgs, void* stream) {
    using namespace ymi;
    YMI_REQUIRE(d != nullptr && imgs != nullptr, "ymi_conv_stem_planar: null argument");
    YMI_REQUIRE(d->w && d->bias && d->y && d->zeros, "ymi_conv_stem_planar: null buffer (w, bias, y and the zero page are required)");
    YMI_REQUIRE(n_imgs == d->n && n_imgs >= 1, "ymi_conv_stem_planar: %d images for a descriptor of batch %d", n_imgs, d->n);
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16, "ymi_conv_stem_planar: dtype must be F16 or BF16");
    YMI_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == YMI_F32, "ymi_conv_stem_planar: out_dtype must equal dtype or be F32");
    YMI_REQUIRE(d->ho == (d->h + 2 * d->ph - d->kh) / d->sh + 1 && d->wo == (d->w_in + 2 * d->pw - d->kw) / d->sw + 1, "ymi_conv_stem_planar: inconsistent output size");
    YMI_REQUIRE(d->y_cstride % 8 == 0 || d->out_dtype == YMI_F32, "ymi_conv_stem_planar: y_cstride must be a multiple of 8");
    for (int i = 0; i < n_imgs; ++i) YMI_REQUIRE(imgs[i] != nullptr && ((uintptr_t)imgs[i] & 15) == 0, "ymi_conv_stem_planar: image %d is null or not 16-byte aligned", i);
    ymi_conv_desc dd = *d;
    dd.x = d->zeros;   // not read; keeps the zero-page offset check of the shared argument builder trivially true
    ConvArgs a;
    { const int rc_args = fill_conv_args(&dd, a); if (rc_args != YMI_OK) return rc_args; }
    return conv_stem_planar_launch(a, imgs, d->dtype, d->out_dtype, (hipStream_t)stream);
}

extern "C" int ymi_conv_head_decode_group(const ymi_conv_desc* convs, int n_levels, const ymi_post_desc* post, void* stream) {
    return ymi::conv_head_decode_group_launch(convs, n_levels, post, (hipStream_t)stream);
}

extern "C" int ymi_conv_head_decode(const ymi_conv_desc* conv, const ymi_post_desc* post, int level, void* stream) {
    return ymi::conv_head_decode_launch(conv, post, level, (hipStream_t)stream);
}

extern "C" int ymi_conv_build_ktab(int cin, int kh, int kw, int w_in, int x_cstride, int k_pad, int32_t* t) {
    if (cin % 8 != 0 || k_pad % 32 != 0 || t == nullptr) {
        ymi::set_error("ymi_conv_build_ktab: cin %% 8 and k_pad %% 32 must be 0");
        return YMI_EINVAL;
    }
    const int K = kh * kw * cin;
    for (int q = 0; q < k_pad / 8; ++q) {
        const int k0 = q * 8;
        if (k0 >= K) {
            t[2 * q] = 0;
            t[2 * q + 1] = -1;
            continue;
        }
        const int tap = k0 / cin, c = k0 - tap * cin;
        const int dy = tap / kw, dx = tap - dy * kw;
        t[2 * q] = (dy * w_in + dx) * x_cstride + c;
        t[2 * q + 1] = (dy << 16) | dx;
    }
    return YMI_OK;
}
