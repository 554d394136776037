// 8-wave implicit-GEMM convolution with LONG steps (gfx950): the general-purpose sibling of conv_halo8.hip for 1x1 and
// strided k x k convolutions (every layer whose cin is a multiple of 32).
//
// Same GEMM view, MFMA mapping, LDS-DMA operand path and epilogues as the 4-wave kernels of conv_igemm_impl.hpp; what
// changes is the schedule (measured on conv_halo8: one barrier per 8 MFMAs per wave leaves both waves of a SIMD in DMA
// issue / ds_read latency at the same time):
//   * block = 8 waves on a 256-pixel x BN-cout tile; one step = BK = 64 k-elements (two k32 sub-stages, each a dense
//     64-byte row image as in v2, XOR-swizzled on the source side) = 16 MFMAs per wave at BN = 128 between barriers;
//   * two-deep ring, plain vmcnt(0): a stage has a whole step (>= 1k cycles) to land;
//   * the 6 DMA pieces a wave issues per step are spread between its four MFMA groups, and the fragments of sub-step
//     i+1 are fetched under the MFMAs of sub-step i; the partner wave of the SIMD covers the rest.
// Out-of-image taps / rows past M read the zero page (no branches); scalar tap arithmetic (cin % 32 == 0: the four chunks
// of a k32 sub-stage share one tap).
//
// Replaces yolort/v5/models/common.py:69-70 (Conv.forward), :115-116 (residual), :172-173 (C3 cv1 + cv2 in one launch).
#include "conv_igemm_impl.hpp"

namespace ymi {

// TP: row-transposed stores (StoreEpilogueTP, conv_igemm_impl.hpp) -- variants 11 .. 19 = tiles 151 .. 159, opt-in
template <int DT, int ODT, int BN, int WAVES_M, int RING, bool TP = false>
__global__ __launch_bounds__(512, RING == 2 ? 2 : 1) void conv_igemm8_kernel(const ConvArgs a) {
    static_assert(RING == 2 || RING == 3, "two- or three-deep stage ring");
    constexpr int WAVES_N = 8 / WAVES_M;
    constexpr int BM = 256;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WAVES_M * WAVES_N == 8, "8 waves");
    constexpr int PA = 2;                              // activation pieces (16 rows) per wave per sub-stage: 16 pieces / 8 waves
    constexpr int W_PIECES = BN / 16;                  // weight pieces per sub-stage (all waves together)
    constexpr int PWS = (W_PIECES + 7) / 8;            // ... per wave
    constexpr int SUB_HALFS = (BM + BN) * 32;          // one k32 sub-stage
    constexpr int STAGE_HALFS = 2 * SUB_HALFS;
    typedef typename Mfma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [stage 0][stage 1]([stage 2])

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = (wave / WAVES_N) * WM, wave_n = (wave % WAVES_N) * WN;

    const int nblk = a.nblk_m * a.nblk_n;
    const int lb = xcd_remap(blockIdx.x, nblk);
    const int bm = lb / a.nblk_n, bn = lb % a.nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk32 = a.k_pad / 32;                     // k32 sub-stages in total
    const int nsteps = (nk32 + 1) / 2;

    // ---- per-lane DMA geometry (as conv_igemm_v2_body, UTAP form) ----
    const int sub_row = lane >> 2;
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
    int a_off[PA], a_mask[PA], a_slot[PA];
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
        unsigned mask = 0;
        int t = 0;
        for (int dy = 0; dy < a.kh; ++dy) {
            const bool yin = (unsigned)(iy0 + dy) < (unsigned)a.h;
            for (int dx = 0; dx < a.kw; ++dx, ++t) {
                const bool in = yin && ((unsigned)(ix0 + dx) < (unsigned)a.w_in);
                mask |= (in ? 1u : 0u) << t;
            }
        }
        a_mask[j] = ok ? (int)mask : 0;
    }
    int w_off[PWS], w_slot[PWS];
#pragma unroll
    for (int j = 0; j < PWS; ++j) {
        int pi = wave * PWS + j;
        pi = pi < W_PIECES ? pi : W_PIECES - 1;
        w_slot[j] = (BM / 16 + pi) * 512;
        // packed weight rows exist up to round_up(cout, 128): a 256-wide tile can reach past them (rows it never stores)
        const int rows_alloc = (a.cout_pad + 127) / 128 * 128;
        int row = n0 + pi * 16 + sub_row;
        row = row < rows_alloc ? row : rows_alloc - 1;
        w_off[j] = row * a.k_pad + chunk * 8;
    }

    // issue-side running position of the next k32 sub-stage (wave-uniform scalars; sub-stages are issued in order)
    int u_kk = 0, u_tap = 0, u_c0 = 0, u_dx = 0, u_kbase = 0;
    auto advance_sub = [&]() {
        ++u_kk;
        u_c0 += 32;
        u_kbase += 32;
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
    };
    // piece p of the CURRENT issue-side sub-stage into `dst` (sub-stage base): p < PA activations, else weights
    auto issue_piece = [&](uint16_t* dst, auto pt) {
        constexpr int p = decltype(pt)::value;
        if constexpr (p < PA) {
            const bool ok = (a_mask[p] >> u_tap) & 1;
            const int off = ok ? a_off[p] + u_kbase + chunk * 8 : a.x_zero_off;
            glds16(a.x + off, dst + a_slot[p]);
        } else {
            glds16(a.w + (w_off[p - PA] + u_kk * 32), dst + w_slot[p - PA]);
        }
    };
    constexpr int PS = PA + PWS;   // pieces per wave per sub-stage

    f32x16 acc[TN][TM];

    // prologue: stage 0 (both sub-stages, or one when K = 32)
    static_for<0, PS>([&](auto pt) { issue_piece(smem, pt); });
    advance_sub();
    if (nk32 > 1) {
        static_for<0, PS>([&](auto pt) { issue_piece(smem + SUB_HALFS, pt); });
        advance_sub();
    }
    f32x4 bias_regs[TN][4];   // issued behind the prologue DMA (conv_common.hpp)
    load_bias<TN>(a, n0 + wave_n, lane >> 5, bias_regs);
    if constexpr (RING == 3) {
        // three-deep ring: stage 1 leaves in the prologue as well, BEHIND the bias loads -- the vector-memory counter retires in
        // order, so init_acc's wait for the bias then covers stage 0 and leaves stage 1 in flight
        if (nk32 > 2) {
            static_for<0, PS>([&](auto pt) { issue_piece(smem + STAGE_HALFS, pt); });
            advance_sub();
        }
        if (nk32 > 3) {
            static_for<0, PS>([&](auto pt) { issue_piece(smem + STAGE_HALFS + SUB_HALFS, pt); });
            advance_sub();
        }
    }
    init_acc<TN, TM>(acc, bias_regs);   // accumulate on top of the bias
    // (after the prologue DMA issue: waiting for the bias load first put two cold memory latencies in series at every block start)

    const int frow = lane & 31;
    const int swz = (lane >> 2) & 3;
    int pos[2];
    pos[0] = ((0 + (lane >> 5)) ^ swz) * 8;
    pos[1] = ((2 + (lane >> 5)) ^ swz) * 8;

    int slot = 0;                              // step % RING
    for (int step = 0; step < nsteps; ++step) {
        if constexpr (RING == 2) {
            wait_vmcnt<0>();                   // the stage was issued one whole step ago
        } else {
            // stage `step` was issued two steps ago; stage step+1 (issued during the previous step, or in the prologue) may stay
            // in flight: exactly the pieces this wave sent for it are newer than everything of stage `step`
            const int n1 = nk32 - 2 * (step + 1);   // k32 sub-stages of stage step+1 (>= 2: a full stage)
            if (n1 >= 2) wait_vmcnt<2 * PS>();
            else if (n1 == 1) wait_vmcnt<PS>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();          // everyone's pieces landed; everyone is done with stage step-1
        __builtin_amdgcn_sched_barrier(0);
        const int subs_here = (2 * step + 1 < nk32) ? 2 : 1;            // k32 sub-stages of this step
        // sub-stages still to issue, for the stage that leaves during this step (step+1, or step+2 with the three-deep ring): >= 2, 1 or <= 0
        const int next_subs = nk32 - 2 * (step + RING - 1);
        const int nslot = RING == 2 ? (slot ^ 1) : (slot == 0 ? 2 : slot - 1);   // (slot + RING - 1) % RING: the stage consumed in step-1
        uint16_t* nstage = smem + nslot * STAGE_HALFS;
        const uint16_t* as = smem + slot * STAGE_HALFS + wave_m * 32;
        const uint16_t* ws = smem + slot * STAGE_HALFS + (BM + wave_n) * 32;
        frag fa[2][TM], fw[2][TN];
        auto read_frags = [&](auto subt, auto buft) {   // sub-step = (k32 sub-stage, k16 half)
            constexpr int sub = decltype(subt)::value, buf = decltype(buft)::value;
            constexpr int ss = sub >> 1, ks = sub & 1;
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[buf][j] = *reinterpret_cast<const frag*>(as + ss * SUB_HALFS + (j * 32 + frow) * 32 + pos[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[buf][i] = *reinterpret_cast<const frag*>(ws + ss * SUB_HALFS + (i * 32 + frow) * 32 + pos[ks]);
        };
        read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, 4>([&](auto subt) {
            constexpr int sub = decltype(subt)::value;
            if constexpr (sub + 1 < 4) {
                if (sub + 1 < 2 * subs_here) read_frags(std::integral_constant<int, sub + 1>{}, std::integral_constant<int, (sub + 1) & 1>{});
            }
            // one slice of the next stage's DMA issue per sub-step: sub-steps 0,1 carry sub-stage 0, sub-steps 2,3 sub-stage 1
            {
                constexpr int half = sub & 1, ss = sub >> 1;
                constexpr int p0 = half * ((PS + 1) / 2), p1 = half ? PS : (PS + 1) / 2;
                if (next_subs > ss) {
                    static_for<p0, p1>([&](auto pt) { issue_piece(nstage + ss * SUB_HALFS, pt); });
                    if constexpr (half == 1) advance_sub();
                }
            }
            if (sub < 2 * subs_here) {
                YMI_PRIO_HI();
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = Mfma<DT>::run(fw[sub & 1][i], fa[sub & 1][j], acc[i][j]);
                YMI_PRIO_LO();
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // done reading this stage before the next barrier
        slot = slot + 1 == RING ? 0 : slot + 1;
    }

    if constexpr (TP) {
        static_assert(RING * 2 * (256 + BN) * 64 >= 8 * LEAN_TP_BYTES<(BN / (8 / WAVES_M)) / 32>, "the operand ring holds the eight waves' store tiles");
        StoreEpilogueTP<DT, ODT>{a}(acc, m0 + wave_m, n0 + wave_n, lane, wave, smem);
    } else {
        StoreEpilogue<DT, ODT>{a}(acc, m0 + wave_m, n0 + wave_n, lane, wave, smem);
    }
}

template <int DT, int ODT, int BN, int WAVES_M, int RING = 2, bool TP = false>
static int launch_igemm8(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.nblk_m = cdiv(a.M, 256);
    a.nblk_n = cdiv(a.cout_pad, BN);
    if (BN == 192 && (a.cout % 192 != 0 || a.split != 0)) {
        set_error("ymi_conv2d: tile 120 (192-cout blocks) needs cout %% 192 == 0 and no channel split");
        return YMI_EINVAL;
    }
    if (a.chain_w != nullptr && !(WAVES_M == 8 && BN == a.chain_k && BN <= 128)) {
        set_error("ymi_conv2d: this tile does not fit the chained 1x1 convolution (pixel-major waves, cout width %d)", a.chain_k);
        return YMI_EINVAL;
    }
    size_t lds = (size_t)RING * 2 * (256 + BN) * 64;
    auto kfn = conv_igemm8_kernel<DT, ODT, BN, WAVES_M, RING, TP>;
    if (lds < lds_floor_bytes()) lds = lds_floor_bytes();
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    hipLaunchKernelGGL(kfn, dim3(a.nblk_m * a.nblk_n), dim3(512), lds, s, a);
    return check_launch("conv_igemm8_kernel");
}

template <int DT, int ODT>
static int igemm8_variant(const ConvArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        case 1: return launch_igemm8<DT, ODT, 128, 4>(a, s);   // 4x2 waves of 64 px x 64 cout
        case 2: return launch_igemm8<DT, ODT, 64, 4>(a, s);    // 4x2 waves of 64 px x 32 cout
        case 3: return launch_igemm8<DT, ODT, 64, 8>(a, s);    // 8x1 waves of 32 px x 64 cout (chained 1x1 with K1 = 64)
        case 4: return launch_igemm8<DT, ODT, 32, 8>(a, s);    // 8x1 waves of 32 px x 32 cout (chained 1x1 with K1 = 32)
        case 5: return launch_igemm8<DT, ODT, 256, 4>(a, s);   // 4x2 waves of 64 px x 128 cout
        case 6: return launch_igemm8<DT, ODT, 128, 8>(a, s);   // 8x1 waves of 32 px x 128 cout
        // three-deep stage ring (one block per CU: 120 / 144 KiB of LDS): two steps of DMA in flight -- the small-M, deep-K layers
        // (one wave of <= 256 blocks, K = 256 .. 1152) spent a full memory latency per step behind the two-deep ring
        case 7: return launch_igemm8<DT, ODT, 128, 4, 3>(a, s);
        case 8: return launch_igemm8<DT, ODT, 128, 8, 3>(a, s);
        case 9: return launch_igemm8<DT, ODT, 64, 4, 3>(a, s);
        // 192-cout blocks (round 4; tile 120): 4x2 waves of 64 px x 96 cout -- yolov5m's 192-cout layers (96 -> 192 and 192 -> 192 stride 2) left a quarter of a 2 x 128 block's MFMAs on zero rows
        case 10: return launch_igemm8<DT, ODT, 192, 4>(a, s);
        default: break;
    }
    if constexpr (ODT == DT) {   // variants 1 / 2 / 5 with row-transposed stores
        switch (variant) {
            case 11: return launch_igemm8<DT, ODT, 128, 4, 2, true>(a, s);
            case 12: return launch_igemm8<DT, ODT, 64, 4, 2, true>(a, s);
            case 15: return launch_igemm8<DT, ODT, 256, 4, 2, true>(a, s);
            default: break;
        }
    }
    {
        set_error("ymi_conv2d: unknown igemm8 variant %d", variant);
        return YMI_EINVAL;
    }
}

int conv_igemm8_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(a.cin % 32 == 0 && a.kh * a.kw <= 32 && a.zeros != nullptr, "ymi_conv2d: the 8-wave implicit-GEMM kernel needs cin %% 32 == 0, <= 32 taps and desc.zeros");
    YMI_REQUIRE(a.k_pad == a.kh * a.kw * a.cin, "ymi_conv2d: igemm8 expects k_pad == kh*kw*cin");
    if (dtype == YMI_F16) return out_dtype == YMI_F32 ? igemm8_variant<YMI_F16, YMI_F32>(a, variant, s) : igemm8_variant<YMI_F16, YMI_F16>(a, variant, s);
    return out_dtype == YMI_F32 ? igemm8_variant<YMI_BF16, YMI_F32>(a, variant, s) : igemm8_variant<YMI_BF16, YMI_BF16>(a, variant, s);
}

}  // namespace ymi
