// fp32 convolution, LDS-DMA pipelined (gfx950) -- the kernel family of the fp32 mode since round 5 (tiles 201-206).
//
// The fp32 mode (fp32 NHWC activations, fp32 folded weights, exact fp32 arithmetic on v_mfma_f32_32x32x2_f32 with two-level
// summation) is the mode in which the HIP path reproduces the reference's fp32 CPU path to rounding-order accuracy (north_star:
// boxes within 1e-3 IoU, equal labels).  Until round 4 it ran on ONE register-staged 128x64 tile with 8-deep steps
// (conv_f32.hip, kept as tile -100: the A/B and bisect partner): 30 % of the 157 TFLOP/s the f32-input MFMA delivers, because a
// step carried 512 cycles of matrix work per wave against a full global-load round trip.  Here:
//   * operands travel HBM/L2 -> LDS with global_load_lds_dwordx4 into a STAGES-deep ring (no VGPR round trip, counted vmcnt,
//     one s_barrier per step); a step is 16 floats deep = 64-byte LDS rows -- BYTE FOR BYTE the piece geometry of the 16-bit
//     kernels (conv_igemm_impl.hpp v2: 1 KiB pieces of 16 rows, XOR swizzle on the source side, conflict-free ds_read_b128);
//   * block tiles up to 128 x 128 (wave tile 64 x 64: 32 MFMAs = 2048 matrix-pipe cycles per wave and step against 8 ds_read_b128
//     and <= 4 DMA instructions), two blocks per CU;
//   * a lane's ds_read_b128 delivers four consecutive k of its row; lanes < 32 take k-chunks {0, 2}, lanes >= 32 chunks {1, 3} of a
//     step.  Two v_permlane32_swap per fragment then hand lanes < 32 the EVEN and lanes >= 32 the ODD k of every pair, so the MFMAs
//     see the k pairs (0, 1), (2, 3), ... in ascending order -- exactly the operands, in exactly the order, of conv_f32.hip;
//   * partial sums of 64 k go to `part` and are folded into `acc` every 4 steps (the summation conv_f32.hip introduced: a plain
//     chain of K FMAs rounds ~sqrt(K) ulp, and the synthetic test networks amplify every ulp on its way to the logits);
//   => every output is BIT-IDENTICAL to the register-staged kernel's, whatever the tile (tests/test_hipsim_kernels.py, tests/test_ops_gpu.py):
//      the detections the reference-made goldens were validated against in rounds 3-4 are reproduced to the last bit, 1.6x faster;
//   * epilogue: + bias, exact SiLU x / (1 + exp(-x)) (torch's CPU formula), + shortcut, 16-byte stores; channel split (C3.cv1 + cv2
//     in one launch) and the x2-upsampled second output (PAN: nn.Upsample folded into its producer) like the 16-bit kernels.
//
// Replaces yolort/v5/models/common.py:69-70 (Conv.forward), :115-116 (Bottleneck shortcut), :172-173 (C3.cv1 / cv2 on one input),
// yolort/models/path_aggregation_network.py:221-223 (1x1 Conv -> nn.Upsample) and yolort/models/box_head.py:36,74 (head conv) -- in
// fp32, like the reference's CPU path.
#include "conv_common.hpp"

namespace ymi {

extern __shared__ __attribute__((aligned(16))) unsigned char f32p_sm[];   // the ONLY LDS object of this unit

constexpr int FBK = 16;   // floats per main-loop step (64-byte rows)

__device__ __forceinline__ float silu_exact_f32(float v) { return v / (1.0f + expf(-v)); }   // torch CPU: x / (1 + exp(-x))

__device__ __forceinline__ void glds16f(const float* g, unsigned char* lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}

// f = {(k0 | k4), (k1 | k5), (k2 | k6), (k3 | k7)} (lanes < 32 | lanes >= 32)  ->  {(k0 | k1), (k4 | k5), (k2 | k3), (k6 | k7)}
__device__ __forceinline__ void pair_even_odd(f32x4& f) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(f[0]), __float_as_uint(f[1]), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(f[2]), __float_as_uint(f[3]), false, false);
    f[0] = __uint_as_float(a[0]);
    f[1] = __uint_as_float(a[1]);
    f[2] = __uint_as_float(b[0]);
    f[3] = __uint_as_float(b[1]);
}

// MODE 0: im2col table (any cin % 8 == 0: the stem's super-pixels), 1: pointwise (k1 s1 p0), 2: uniform tap (cin % 16 == 0: a step lies inside one tap)
// What was tried on top of this loop in round 5 and measured on every convolution of the yolov5s bs-32 plan (profiles/r05c_f32_tile_sweep_pipelined_c2.txt,
// r05d_f32_tile_experiments_c2.txt, r05d_f32_one_block_per_cu_c2.txt) -- all bit-identical, none kept:
//   * a software-pipelined loop (next half-step's ds_read_b128 and the ring refill issued between the MFMAs): +-1 % on every layer -- the 64-cycle f32 MFMA leaves the
//     issue port idle anyway;
//   * a four-deep ring: +-1 % (the loop is not DMA-latency-bound);
//   * single-wave blocks of 64 x 64 (no s_barrier at all): 1.2-1.3 x SLOWER (twice the L2 -> LDS operand traffic per flop);
//   * one block per CU (LDS floor): 1.2-1.7 x slower -- a single block reaches ~0.6 of its matrix-pipe time, the second resident block fills the rest.
// SQ_VALU_MFMA_BUSY_CYCLES is 0.53 of the SIMD-cycles at an effective clock of 2.3 GHz (profiles/r05b_rocprof_summary_c2_fp32.csv): the remainder is block-count
// quantisation over 256 CUs (400-800 blocks per launch at 40 x 40 / 20 x 20), the LDS-DMA issue cost inside the MFMA stream and the exact-SiLU epilogue.
template <int BM, int BN, int WM, int WN, int STAGES, int MODE>
__global__ __launch_bounds__(256, 2) void conv_f32_pipe_kernel(const ConvArgs a) {
    constexpr int NW = 4;
    static_assert((BM / WM) * (BN / WN) == NW, "4 waves per block");
    static_assert(BM % 16 == 0 && BN % 16 == 0 && (BM / 16) % NW == 0, "activation pieces are dealt evenly to the waves");
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int PA = BM / 16 / NW;             // activation pieces (1 KiB = 16 rows x 64 B) per wave per stage
    constexpr int W_PIECES = BN / 16;
    constexpr int PW = (W_PIECES + NW - 1) / NW; // weight pieces per wave per stage (surplus waves re-send the last piece: identical bytes)
    constexpr int P = PA + PW;                   // DMA instructions per wave per stage, the same for every wave
    constexpr int STAGE_BYTES = (BM + BN) * 64;

    const float* __restrict__ X = reinterpret_cast<const float*>(a.x);
    const float* __restrict__ Wt = reinterpret_cast<const float*>(a.w);
    int2* ktab_lds = reinterpret_cast<int2*>(f32p_sm + STAGES * STAGE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = (wave / WAVES_N) * WM, wave_n = (wave % WAVES_N) * WN;
    const int hi = lane >> 5, frow = lane & 31;

    const int nblk = a.nblk_m * a.nblk_n;
    const int lb = xcd_remap(blockIdx.x, nblk);
    const int bm = lb / a.nblk_n, bn = lb % a.nblk_n;   // cout tile fastest: the blocks sharing an activation tile run back to back on one XCD
    const int m0 = bm * BM, n0 = bn * BN;
    const int nsteps = a.k_pad / FBK;

    // bias first: these loads retire before every operand DMA issued behind them (in-order return), so the counted waits below stay exact
    f32x4 bias_regs[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = n0 + wave_n + i * 32 + g * 8 + hi * 4;
            bias_regs[i][g] = *reinterpret_cast<const f32x4*>(a.bias + (co < a.cout_pad ? co : 0));   // rows past cout_pad are never stored
        }

    if constexpr (MODE == 0) {
        for (int i = tid; i < a.k_pad / 8; i += 64 * NW) ktab_lds[i] = a.ktab[i];
    }

    // ---- per-lane DMA geometry (element = float offsets against a.x / a.w; out-of-range chunks read the zero page in x's own tail) ----
    const int sub_row = lane >> 2;
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);      // k-chunk fetched = position ^ ((row >> 2) & 3): linear destination, permuted source
    int a_off[PA], a_aux[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int pi = wave * PA + j;
        const int m = m0 + pi * 16 + sub_row;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        if constexpr (MODE == 1) {
            a_off[j] = mm * a.x_cs;
            a_aux[j] = ok ? 0 : -1;
        } else {
            const int hw_o = a.ho * a.wo;
            const int img = fast_div(mm, hw_o, a.magic_hw);
            const int rem = mm - img * hw_o;
            const int oy = fast_div(rem, a.wo, a.magic_w), ox = rem - oy * a.wo;
            const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
            a_off[j] = ((img * a.h + iy0) * a.w_in + ix0) * a.x_cs;
            if constexpr (MODE == 2) {
                unsigned mask = 0;
                int t = 0;
                for (int dy = 0; dy < a.kh; ++dy) {
                    const bool yin = (unsigned)(iy0 + dy) < (unsigned)a.h;
                    for (int dx = 0; dx < a.kw; ++dx, ++t) mask |= ((yin && ((unsigned)(ix0 + dx) < (unsigned)a.w_in)) ? 1u : 0u) << t;
                }
                a_aux[j] = ok ? (int)mask : 0;
            } else {
                a_aux[j] = ok ? (((iy0 + 16384) << 16) | ((ix0 + 16384) & 0xffff)) : -1;
            }
        }
    }
    int w_off[PW], w_slot[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        int pi = wave * PW + j;
        pi = pi < W_PIECES ? pi : W_PIECES - 1;
        w_slot[j] = (BM / 16 + pi) * 1024;
        w_off[j] = (n0 + pi * 16 + sub_row) * a.k_pad + chunk * 4;   // rows are zero-padded to a multiple of 128 (engine.PackedConv)
    }

    int u_tap = 0, u_c0 = 0, u_dx = 0, u_kbase = 0;   // MODE 2 running state (stages are issued in order): wave-uniform scalars
    // one stage = issue_begin(step); issue_piece(0 .. P-1); issue_end()   (pieces 0 .. PA-1 activations, PA .. P-1 weights)
    unsigned char* cur_stage = f32p_sm;
    int cur_koff = 0, cur_dy = 0, cur_dx = 0, cur_step = 0;
    bool cur_tap_ok = true;
    auto issue_begin = [&](int step) {
        cur_stage = f32p_sm + (step % STAGES) * STAGE_BYTES;
        cur_step = step;
        if constexpr (MODE == 2) {
            cur_koff = u_kbase + chunk * 4;
        } else if constexpr (MODE == 1) {
            cur_koff = step * FBK + chunk * 4;
            cur_tap_ok = cur_koff < a.cin;
        } else {
            const int2 t = ktab_lds[step * 2 + (chunk >> 1)];   // one table entry per 8 elements: two 4-float chunks
            cur_koff = t.x + (chunk & 1) * 4;
            cur_tap_ok = t.y >= 0;
            cur_dy = t.y >> 16;
            cur_dx = t.y & 0xffff;
        }
    };
    auto issue_piece = [&](auto jt) {
        constexpr int j = decltype(jt)::value;
        if constexpr (j < PA) {
            bool ok;
            if constexpr (MODE == 2) {
                ok = (a_aux[j] >> u_tap) & 1;
            } else {
                ok = cur_tap_ok & (a_aux[j] >= 0);
                if constexpr (MODE == 0) {
                    const int iy = (a_aux[j] >> 16) - 16384 + cur_dy, ix = (a_aux[j] & 0xffff) - 16384 + cur_dx;
                    ok = ok & ((unsigned)iy < (unsigned)a.h) & ((unsigned)ix < (unsigned)a.w_in);
                }
            }
            glds16f(X + (ok ? a_off[j] + cur_koff : a.x_zero_off), cur_stage + (wave * PA + j) * 1024);
        } else if constexpr (j < P) {
            glds16f(Wt + (w_off[j - PA] + cur_step * FBK), cur_stage + w_slot[j - PA]);
        }
    };
    auto issue_end = [&]() {
        if constexpr (MODE == 2) {
            u_c0 += FBK;
            u_kbase += FBK;
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

    f32x16 acc[TN][TM], part[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; part[i][j][r] = 0.f; }

    if constexpr (MODE == 0) __syncthreads();   // table visible (no DMA in flight yet)
    const int swz = (lane >> 2) & 3;
    const int pos0 = ((0 + hi) ^ swz) * 16, pos1 = ((2 + hi) ^ swz) * 16;   // byte offset of this lane's k-chunk within its row, ks = 0 / 1
    auto fold = [&](int step) {
        if ((step & 3) == 3 || step + 1 == nsteps) {   // 64 k per partial sum
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[i][j][r] += part[i][j][r]; part[i][j][r] = 0.f; }
        }
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) issue(s);

    for (int step = 0; step < nsteps; ++step) {
        // this wave's pieces of stage `step` have landed once at most `ahead` later stages are pending
        const int issued = (step + STAGES - 1 < nsteps) ? step + STAGES - 1 : nsteps;
        const int ahead = issued - (step + 1);
        if (ahead >= 2) wait_vmcnt<2 * P>();
        else if (ahead == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();          // every wave's pieces landed; everyone is done with stage step - 1
        __builtin_amdgcn_sched_barrier(0);
        if (step + STAGES - 1 < nsteps) issue(step + STAGES - 1);   // refill the slot freed by step - 1
        const unsigned char* as = f32p_sm + (step % STAGES) * STAGE_BYTES + wave_m * 64;
        const unsigned char* ws = f32p_sm + (step % STAGES) * STAGE_BYTES + (BM + wave_n) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pos = ks == 0 ? pos0 : pos1;
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const f32x4*>(as + (j * 32 + frow) * 64 + pos);
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f32x4*>(ws + (i * 32 + frow) * 64 + pos);
            // (k | k + 4) pairs -> (k | k + 1) pairs: component 0 <-> 1 and 2 <-> 3 exchange their upper / lower lane halves
#pragma unroll
            for (int j = 0; j < TM; ++j) pair_even_odd(af[j]);
#pragma unroll
            for (int i = 0; i < TN; ++i) pair_even_odd(wf[i]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = (q & 1) * 2 + (q >> 1);   // components 0, 2, 1, 3 hold the k pairs (0, 1), (2, 3), (4, 5), (6, 7) of this half-step
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i][e], af[j][e], part[i][j], 0, 0, 0);
            }
        }
        fold(step);
    }

    // ---- epilogue: + bias, exact SiLU, + shortcut (after the activation), 16-byte fp32 stores ----
    const float* __restrict__ R = reinterpret_cast<const float*>(a.res);
    const int hw_o = a.ho * a.wo;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int64_t mo = (int64_t)m0 + wave_m + j * 32 + frow;
        if (mo >= a.M) continue;
        int64_t m_up = 0;
        if (a.up2) {   // wave-uniform flag: pixel (img, oy, ox) -> top-left of its 2 x 2 block in the (n, 2 ho, 2 wo) view y2
            const int img = fast_div((int)mo, hw_o, a.magic_hw);
            const int rem = (int)mo - img * hw_o;
            const int oy = fast_div(rem, a.wo, a.magic_w), ox = rem - oy * a.wo;
            m_up = ((int64_t)img * 2 * a.ho + 2 * oy) * (2 * a.wo) + 2 * ox;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = n0 + wave_n + i * 32 + g * 8 + hi * 4;
                if (co >= a.cout) continue;
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                const bool full = co + 3 < a.cout;
                if (R != nullptr) {
                    if (full && (a.res_cs & 3) == 0) rr = *reinterpret_cast<const f32x4*>(R + mo * a.res_cs + co);
                    else
                        for (int e = 0; e < 4 && co + e < a.cout; ++e) rr[e] = R[mo * a.res_cs + co + e];
                }
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][g * 4 + e] + bias_regs[i][g][e];
                    if (a.act == YMI_ACT_SILU) t = silu_exact_f32(t);
                    if (R != nullptr) t += rr[e];
                    v[e] = t;
                }
                float* yp;
                int cs;
                if (a.split > 0 && co >= a.split) { yp = reinterpret_cast<float*>(a.y2) + mo * a.y2_cs + (co - a.split); cs = a.y2_cs; }
                else { yp = reinterpret_cast<float*>(a.y) + mo * a.y_cs + co; cs = a.y_cs; }
                if (full && (cs & 3) == 0 && (a.split & 3) == 0) {
                    *reinterpret_cast<f32x4*>(yp) = v;
                } else {
                    for (int e = 0; e < 4 && co + e < a.cout; ++e) yp[e] = v[e];
                }
                if (a.up2) {   // launch checks: cout % 4 == 0, y2_cs % 4 == 0, no split
                    float* up = reinterpret_cast<float*>(a.y2) + m_up * a.y2_cs + co;
                    const int64_t row = (int64_t)2 * a.wo * a.y2_cs;
                    *reinterpret_cast<f32x4*>(up) = v;
                    *reinterpret_cast<f32x4*>(up + a.y2_cs) = v;
                    *reinterpret_cast<f32x4*>(up + row) = v;
                    *reinterpret_cast<f32x4*>(up + row + a.y2_cs) = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int STAGES>
static int launch_f32_pipe(const ConvArgs& a0, bool is1x1, hipStream_t s) {
    ConvArgs a = a0;
    a.nblk_m = cdiv(a.M, BM);
    a.nblk_n = cdiv(a.cout_pad, BN);
    const bool utap = !is1x1 && (a.cin % FBK == 0) && (a.kh * a.kw <= 32);
    size_t lds = (size_t)STAGES * (BM + BN) * 64 + ((is1x1 || utap) ? 0 : (size_t)a.k_pad) + 16;
    if (lds < lds_floor_bytes()) lds = lds_floor_bytes();
    dim3 grid(a.nblk_m * a.nblk_n), block(256);
    auto go = [&](auto kfn) -> int {
        if (lds > 64 * 1024) { const int rc = allow_big_lds((const void*)kfn, (int)lds); if (rc != YMI_OK) return rc; }
        hipLaunchKernelGGL(kfn, grid, block, lds, s, a);
        return check_launch("conv_f32_pipe_kernel");
    };
    if (is1x1) return go(conv_f32_pipe_kernel<BM, BN, WM, WN, STAGES, 1>);
    if (utap) return go(conv_f32_pipe_kernel<BM, BN, WM, WN, STAGES, 2>);
    return go(conv_f32_pipe_kernel<BM, BN, WM, WN, STAGES, 0>);
}

// Tile choice for tile id 0: a function of the shape only.
// Fitted to the measurement of every tile on every convolution of the yolov5s bs-32 plan (profiles/r05a_f32_tile_sweep_c2.txt): 128 x 64 (four waves per SIMD resident,
// twice the blocks of 128 x 128 to balance over 256 CUs) wins or ties within 3 % wherever cout >= 64, except for the 512-wide layers (128 x 128: 3-5 %); the 32-cout
// layers take 128 x 32.  Every tile produces the same bits, so the choice is a matter of time only.
int conv_f32_pick_tile(int M, int cout_pad) {
    if (cout_pad <= 32) return 206;
    if (cout_pad >= 512 && (long)cdiv(M, 128) * cdiv(cout_pad, 128) >= 256) return 201;
    return 202;
}

int conv_f32_pipe_launch(const ConvArgs& a0, bool is1x1, int tile, hipStream_t s) {
    ConvArgs a = a0;
    YMI_REQUIRE(a.zeros != nullptr, "ymi_conv2d (fp32, pipelined tiles): desc.zeros is required (the 256-byte zero tail of x's buffer)");
    YMI_REQUIRE(a.chain_w == nullptr, "ymi_conv2d: the fp32 mode has no chained convolution");
    YMI_REQUIRE(a.x_cs % 4 == 0 && a.cin % 8 == 0 && a.k_pad % 32 == 0, "ymi_conv2d (fp32): x_cstride %% 4, cin %% 8 and k_pad %% 32 must be 0");
    YMI_REQUIRE(a.split == 0 || a.split % 4 == 0, "ymi_conv2d (fp32): cout_split must be a multiple of 4");
    YMI_REQUIRE(!a.up2 || (a.cout % 4 == 0 && a.y2_cs % 4 == 0 && a.split == 0 && a.y_cs % 4 == 0), "ymi_conv2d (fp32): the upsampled second output needs cout %% 4 == 0 and 16-byte aligned pixel strides");
    const int64_t dz = ((const char*)a.zeros - (const char*)a.x);
    YMI_REQUIRE(dz % 16 == 0 && dz / 4 > -((int64_t)1 << 31) && dz / 4 < ((int64_t)1 << 31), "ymi_conv2d (fp32): desc.zeros must lie within range of x and be 16-byte aligned relative to it");
    a.x_zero_off = (int)(dz / 4);
    YMI_REQUIRE((int64_t)a.n * a.h * a.w_in * a.x_cs < ((int64_t)1 << 31) && (int64_t)(a.cout_pad + 127) / 128 * 128 * a.k_pad < ((int64_t)1 << 31), "ymi_conv2d (fp32): tensor too large for 32-bit element offsets");
    if (tile == 0) tile = conv_f32_pick_tile(a.M, a.cout_pad);
    switch (tile) {
        case 201: return launch_f32_pipe<128, 128, 64, 64, 3>(a, is1x1, s);
        case 202: return launch_f32_pipe<128, 64, 64, 32, 3>(a, is1x1, s);
        case 203: return launch_f32_pipe<64, 64, 32, 32, 3>(a, is1x1, s);
        case 204: return launch_f32_pipe<256, 64, 64, 64, 3>(a, is1x1, s);
        case 205: return launch_f32_pipe<256, 32, 64, 32, 3>(a, is1x1, s);
        case 206: return launch_f32_pipe<128, 32, 32, 32, 3>(a, is1x1, s);
        default: break;
    }
    set_error("ymi_conv2d (fp32): unknown tile id %d (201-206, 0 = by shape, negative = the register-staged kernel)", tile);
    return YMI_EINVAL;
}

}  // namespace ymi
