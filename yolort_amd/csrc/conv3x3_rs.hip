// ROW-STREAMING 3x3 convolution of a 64-channel input, weights in registers (gfx950; round 4): tile 137 (stride 1, cout = 64) and tile 138 (stride 2, cout = 128).
//
// What round 4 measured on the patch-tiled resident-weights kernels (tiles 133 / 134; profiles/r04h_tile134_ablation.txt, r04j_lds_mfma_bench.txt): a block that stages
// one 17 x 17 input patch per 8 x 8 output tile has, per CU, two patches (78 KiB) of LDS but on average little more than one of them IN FLIGHT, and at the ~4 us the memory
// system takes to answer under load that is ~3.2 TB/s whatever is done to the compute loop, the stores or the number of buffers.  The patch form also re-reads the halo of
// every tile (13 %) and restarts its pipeline per tile.  Here a block walks DOWN a 16-column strip of the output and keeps a ring of input ROWS in LDS:
//   * work = a contiguous range of STEPS of one strip; a step produces TH output rows x 16 columns (stride 2: TH = 2, all four waves share the 32 pixels and own one
//     32-cout group each; stride 1: TH = 4, two pixel groups x two cout groups) from a window of S*TH + 2 ... input rows and advances the stream by FOUR rows;
//   * the images of a strip are concatenated into one stream of rows -- per image one zero row (the top padding), its H rows, and zero rows up to a multiple of four -- so
//     the pipeline never drains between images; the one or two output rows per image that straddle the padding are computed and not stored;
//   * rows arrive in GROUPS of four by LDS-DMA, D groups ahead of the step that needs them, into a ring of R rows; no row is fetched twice vertically, the horizontal halo is
//     two columns per strip (6 % at stride 2, 12 % at stride 1);
//   * ONE barrier per step; each wave waits for its own DMA pieces with a COUNTED s_waitcnt -- "at most as many operations as the D - 1 younger groups have pieces are
//     outstanding" -- so the row groups ahead stay in flight across the barrier.  The output stores share the counter; the count does not rely on them (loads complete in
//     order among loads, stores among stores, nothing is assumed between the two kinds -- see the comment at the wait);
//   * fragment reads are inline assembly, PF units ahead of their MFMA, with counted lgkmcnt: left to the compiler this loop pays the full LDS latency
//     before every MFMA.
// Row slots are 128 bytes per pixel; at stride 2 a row keeps its columns split by parity ([even | odd]); chunk swizzle v = (column index >> 1) & 7 on top of the slot parity:
// a ds_read_b128 lane group (lanes {0-3, 12-15} of one tile row and {20-27} of the next) touches 16 different 16-byte bank units.
// K order (tap-major, channel-minor), single accumulator chain, lean epilogue: bit-identical to the implicit GEMM (tiles 111-113).
// Replaces yolort/v5/models/common.py:69-70 for Conv(64, 128, 3, 2) and Bottleneck(64, 64).cv2 (its shortcut: :115-116 -- not fused here yet, the launcher refuses `res`).
#include "conv_common.hpp"
#include <cstdlib>

namespace ymi {

#ifndef YMI_LDS_ASM_HELPERS
#define YMI_LDS_ASM_HELPERS
template <class F>
__device__ __forceinline__ void rs_lds_read16(F& dst, const unsigned char* p, unsigned lds_addr) {
#ifdef YMI_HIPSIM
    (void)lds_addr;
    dst = *reinterpret_cast<const F*>(p);
#else
    (void)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr));
#endif
}
template <int N>
__device__ __forceinline__ void rs_lds_wait() {
#ifndef YMI_HIPSIM
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int N>
__device__ __forceinline__ void rs_vm_wait() {
#ifndef YMI_HIPSIM
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ unsigned rs_lds_addr(const unsigned char* p) {
#ifdef YMI_HIPSIM
    return 0u;
#else
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
#endif
}
#endif

constexpr int RS_TW = 16;        // output columns of a strip
constexpr int RS_BIAS_BYTES = 2048;   // bias quads (<= 512 B) + a 1 KiB landing area for the count-keeping dummy loads (rs_sm + 1024)

template <int S>
struct RsCfg {
    static constexpr int NG = S == 2 ? 4 : 2;              // 32-cout groups: cout = 128 (stride 2) / 64 (stride 1)
    static constexpr int NPG = 4 / NG;                     // pixel groups per step (a pixel group = 2 output rows x 16 columns = one MFMA pixel tile)
    static constexpr int TH = 2 * NPG;                     // output rows per step
    static constexpr int WIN = S * (TH - 1) + 3;           // stream rows a step reads: 5 (stride 2) / 6 (stride 1); it advances by S * TH = 4
    static constexpr int SLOTS = S == 2 ? 36 : 18;         // 128-byte pixel slots per row: stride 2: even columns (index 0..16) at 0..16, odd ones at 18..34; stride 1: 18 columns
    static constexpr int HO = 18;                          // stride 2: first slot of the odd columns
    static constexpr int PXW = S * RS_TW + 2;              // input columns a strip needs: 34 / 18
    static constexpr int ROWB = SLOTS * 128;               // 4608 / 2304 bytes: a multiple of the 256-byte bank row
    static constexpr int R = S == 2 ? 16 : 32;             // ring rows (a multiple of four): 72 KiB per block, two blocks per CU
    static constexpr int D = S == 2 ? 2 : 6;               // groups in flight ahead of the two a step reads: R / 4 >= D + 2
    static constexpr int GPIECES = 4 * SLOTS * 8 / 64;     // DMA pieces (1 KiB) per group of four rows: 18 / 9
    static constexpr int PPW = (GPIECES + 3) / 4;          // ... per wave, at most: 5 / 3 (waves with fewer issue exactly their share: the counted waits know)
    static constexpr int PF = 6;                           // fragment units in flight
};

template <int DT, int S, bool COUNTED>
__global__ __launch_bounds__(256, 2) void conv3x3_rs_kernel(const ConvArgs a, int strips, int steps_per_image, int chunk_steps, int chunks_per_strip) {
    typedef typename Mfma<DT>::frag frag;
    typedef RsCfg<S> C;
    constexpr int KC = 4, NU = 9 * KC;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_sm[];
    f32x4* bl = reinterpret_cast<f32x4*>(rs_sm);
    unsigned char* ring = rs_sm + RS_BIAS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;
    const int ct = wave % C::NG, pg = wave / C::NG;        // cout group, pixel group of the step
    // (stream rows per image: 1 zero row + H rows, rounded up to a multiple of four = 4 * steps_per_image)

    // ---- work: a contiguous range of steps of one strip ----
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = item / chunks_per_strip, chunk = item - strip * chunks_per_strip;
    const int steps_total = a.n * steps_per_image;
    const int s0 = chunk * chunk_steps;
    const int s1 = s0 + chunk_steps < steps_total ? s0 + chunk_steps : steps_total;
    if (strip >= strips || s0 >= s1) return;                // (whole blocks only: no barrier has been passed)
    const int x0 = strip * RS_TW;                           // first output column
    const int ix0 = S * x0 - 1;                             // first input column of the row slots

    // ---- this wave's weights: fragment (tap, kc) = rows ct*32 + frow, k = tap*64 + kc*16 + hi*8 .. +7 ----
    frag wf[NU];
    {
        const uint16_t* wr = a.w + (int64_t)(ct * 32 + frow) * a.k_pad + hi * 8;
#pragma unroll
        for (int u = 0; u < NU; ++u) wf[u] = *reinterpret_cast<const frag*>(wr + u * 16);
    }
    if (tid < 8 * C::NG) {   // bias quad of (group t, octet g, half h): couts t*32 + g*8 + h*4 ..
        const int t = tid >> 3, g = (tid >> 1) & 3, h = tid & 1;
        bl[tid] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + g * 8 + h * 4);
    }

    // ---- DMA geometry of this wave's pieces of a row group (fixed per lane): piece pi = wave + 4 j; entry e = pi*64 + lane of the group's 4 * ROWB contiguous bytes ----
    int p_rc[C::PPW];    // row in group << 16 | chunk << 8 | input column relative to ix0, or -1: a pad slot (nothing to fetch)
#pragma unroll
    for (int j = 0; j < C::PPW; ++j) {
        const int pi = wave + 4 * j;
        const int e = pi * 64 + lane;
        const int q = e >> 3;                              // slot index within the group
        const int rr = q / C::SLOTS, sc = q - rr * C::SLOTS;
        int col, ci;
        if (S == 2) { const int odd = sc >= C::HO ? 1 : 0; ci = odd ? sc - C::HO : sc; col = 2 * ci + odd; if (ci > RS_TW) col = 1 << 20; }
        else { ci = sc; col = sc; }
        const int chunk_ = (e & 7) ^ ((ci >> 1) & 7);
        p_rc[j] = (pi < C::GPIECES && col < C::PXW) ? ((rr << 16) | (chunk_ << 8) | col) : -1;
    }
    const int npw = (C::GPIECES - wave + 3) / 4;           // pieces this wave really issues per group (wave-uniform): 5 5 4 4 / 3 2 2 2
    auto issue_group = [&](int g) {   // stream rows 4g .. 4g+3 of this strip -> ring rows (4g) % R ..
        const int img = g / steps_per_image;               // (LB is a multiple of four: a group never straddles two images)
        const int jj0 = 4 * (g - img * steps_per_image);    // stream row within the image block: 0 = the zero row, 1 .. H = input rows 0 .. H-1, then zero rows
        const bool live = img < a.n;
        unsigned char* dst = ring + ((4 * g) % C::R) * C::ROWB;
#pragma unroll
        for (int j = 0; j < C::PPW; ++j) {
            if (j < npw) {
                const int pi = wave + 4 * j;
                const int iy = jj0 + (p_rc[j] >> 16) - 1, ix = ix0 + (p_rc[j] & 0xff);
                const bool ok = live && p_rc[j] >= 0 && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
                const int off = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + ((p_rc[j] >> 8) & 0xff) * 8 : a.x_zero_off;
                glds16(a.x + off, reinterpret_cast<uint16_t*>(dst + pi * 1024));
            }
        }
    };

    // ---- fragment geometry (fixed per lane): pixel (r, c) = (frow >> 4, frow & 15) of pixel group pg -> output row TH-local 2 pg + r ----
    const int pr_l = 2 * pg + (frow >> 4), pc_l = frow & 15;
    int eb[3];            // byte offset within a row of chunk `hi` (k16 step 0) of tap column dx; step kc: ^ (kc << 5)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int col = S * pc_l + dx;
        const int ci = S == 2 ? (col >> 1) : col;
        const int slot = S == 2 ? (((col & 1) ? C::HO : 0) + ci) : col;
        eb[dx] = slot * 128 + ((hi ^ ((ci >> 1) & 7)) * 16);
    }
    const unsigned ring_lds = rs_lds_addr(ring);

    // ---- prologue: the groups the first step reads (s0, s0 + 1) and D - 1 ... more: groups s0 .. s0 + D ----
#pragma unroll 1
    for (int g = s0; g <= s0 + C::D; ++g) issue_group(g);

    for (int s = s0; s < s1; ++s) {
        // groups <= s + 1 must have landed.  Younger LOADS: the D - 1 groups s + 2 .. s + D.  The wave's output STORES of the last steps sit on the same counter, and
        // gfx9 orders vector-memory completions only within a kind (loads among loads, stores among stores; LLVM's waitcnt pass treats a counter holding both kinds as
        // out of order for that reason).  So the count must not ASSUME pending stores: with N = (younger loads) the wait passes only when at most N operations of any
        // kind are pending, hence at most N loads, hence -- loads retiring in order -- the group-(s + 1) pieces have landed whatever the stores did.  (Round 4 added
        // 2 D for "the stores of the last D steps": correct only if stores never overtake older loads -- ADVICE r4; the strict count costs nothing measurable.)
        if (COUNTED) {
            if (npw == C::PPW) rs_vm_wait<(C::D - 1) * C::PPW>(); else rs_vm_wait<(C::D - 1) * (C::PPW - 1)>();
        } else {
            rs_vm_wait<0>();
        }
        __builtin_amdgcn_s_barrier();   // (a raw barrier: __syncthreads() would drain the vector-memory counter and with it the row groups in flight)
        issue_group(s + C::D + 1);      // into the ring rows of group s - 1 ... (R / 4 >= D + 2): everyone has left them

        const int img = s / steps_per_image, ls = s - img * steps_per_image;
        const int b0 = (4 * s) % C::R;  // ring row of the window's first stream row
        unsigned rowb[3];               // LDS byte address of this lane's window row dy
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            int ro = b0 + S * pr_l + dy;
            ro -= ro >= C::R ? C::R : 0;
            rowb[dy] = (unsigned)(ro * C::ROWB);
        }
        f32x16 acc[1][1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = bl[(ct * 4 + g) * 2 + hi];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][0][g * 4 + e] = b[e];
        }
        frag fa[C::PF];
        auto read_unit = [&](auto ut) {
            constexpr int u = decltype(ut)::value;
            constexpr int t = u / KC, kc = u % KC, dy = t / 3, dx = t % 3;
            const unsigned off = rowb[dy] + (unsigned)(eb[dx] ^ (kc << 5));
            rs_lds_read16(fa[u % C::PF], ring + off, ring_lds + off);
        };
        static_for<0, C::PF - 1>([&](auto ut) { read_unit(ut); });
        static_for<0, NU>([&](auto ut) {
            constexpr int u = decltype(ut)::value;
            if constexpr (u + C::PF - 1 < NU) read_unit(std::integral_constant<int, u + C::PF - 1>{});
            constexpr int younger = (u + C::PF - 1 < NU ? C::PF - 1 : NU - 1 - u);
            rs_lds_wait<younger>();
            acc[0][0] = Mfma<DT>::run(wf[u], fa[u % C::PF], acc[0][0]);
        });
        // lean epilogue: SiLU, 16-byte packets; rows past the image (the padding of the stream) and columns past the map are not stored
        const int oy = ls * C::TH + pr_l, ox = x0 + pc_l;
        auto pix = [&](int, int64_t& m, bool& ok) {
            ok = oy < a.ho && ox < a.wo;
            m = ((int64_t)img * a.ho + oy) * a.wo + ox;
        };
        finish_wave_tile_lean<DT, 1, 1, false>(a, acc, ct * 32, hi, pix);
        if (COUNTED) {
            // the counted wait above assumes TWO output stores per wave and step.  A wave whose two output rows both lie in the padding of the stream (the last step of an
            // image) issues none (every lane is masked off): it sends two one-lane loads of the zero page into the bias block's spare bytes instead, so that the count holds
            if (ls * C::TH + 2 * pg >= a.ho) {
                glds16(a.x + a.x_zero_off, reinterpret_cast<uint16_t*>(rs_sm + 1024));
                glds16(a.x + a.x_zero_off, reinterpret_cast<uint16_t*>(rs_sm + 1024));
            }
        }
    }
}

template <int DT, int S>
static int launch_rs(const ConvArgs& a0, hipStream_t s) {
    typedef RsCfg<S> C;
    ConvArgs a = a0;
    const int strips = cdiv(a.wo, RS_TW);
    const int steps_per_image = cdiv(a.h + 1, 4);                       // stream rows per image = 4 * this >= H + 1; output rows 0 .. TH * this - 1 >= Ho
    const int steps_total = a.n * steps_per_image;
    int resident = 512;                                                 // two 4-wave blocks per CU
    if (const char* e = getenv("YOLORT_AMD_RES3X3_BLOCKS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4096) resident = v;
    }
    int chunks_per_strip = cdiv(resident, strips);
    if (chunks_per_strip > steps_total) chunks_per_strip = steps_total;
    const int chunk_steps = cdiv(steps_total, chunks_per_strip);
    chunks_per_strip = cdiv(steps_total, chunk_steps);                  // no empty chunks
    const size_t lds = RS_BIAS_BYTES + (size_t)C::R * C::ROWB;
    const char* counted_env = getenv("YOLORT_AMD_RS_COUNTED");                  // A/B and stress-test knob, read per launch: 0 = drain everything at every step
    const bool counted = counted_env == nullptr || atoi(counted_env) != 0;
    a.nblk_m = strips * chunks_per_strip;
    a.nblk_n = 1;
    if (counted) {
        auto kfn = conv3x3_rs_kernel<DT, S, true>;
        if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
        hipLaunchKernelGGL(kfn, dim3(strips * chunks_per_strip), dim3(256), lds, s, a, strips, steps_per_image, chunk_steps, chunks_per_strip);
    } else {
        auto kfn = conv3x3_rs_kernel<DT, S, false>;
        if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
        hipLaunchKernelGGL(kfn, dim3(strips * chunks_per_strip), dim3(256), lds, s, a, strips, steps_per_image, chunk_steps, chunks_per_strip);
    }
    return check_launch("conv3x3_rs_kernel");
}

// variant 1: stride 1, 64 -> 64 (tile 137); variant 2: stride 2, 64 -> 128 (tile 138)
int conv3x3_rs_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(variant == 1 || variant == 2, "ymi_conv2d: unknown row-streaming 3x3 variant %d", variant);
    const int S = variant;
    YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.ph == 1 && a.pw == 1 && a.sh == S && a.sw == S && a.cin == 64 && a.k_pad >= 576 && a.cout == (S == 2 ? 128 : 64) && a.cout_pad >= a.cout &&
                    a.zeros != nullptr && a.up2 == 0 && a.split == 0 && a.chain_w == nullptr && a.res == nullptr && out_dtype == dtype && a.act == YMI_ACT_SILU,
                "ymi_conv2d: the row-streaming 3x3 kernel (tiles 137 / 138) handles cin = 64, cout = 64 (stride 1) / 128 (stride 2), pad 1, SiLU, 16-bit output, no shortcut / chained conv");
    YMI_REQUIRE(a.ho == (a.h + 2 - 3) / S + 1 && a.wo == (a.w_in + 2 - 3) / S + 1, "ymi_conv2d: tiles 137 / 138: inconsistent output size");
    YMI_REQUIRE(((int64_t)a.M + 1) * a.y_cs < ((int64_t)1 << 31) && (int64_t)a.n * a.h * a.w_in * a.x_cs < ((int64_t)1 << 31), "ymi_conv2d: tiles 137 / 138: tensor too large for 32-bit offsets");
    if (S == 1) return dtype == YMI_F16 ? launch_rs<YMI_F16, 1>(a, s) : launch_rs<YMI_BF16, 1>(a, s);
    return dtype == YMI_F16 ? launch_rs<YMI_F16, 2>(a, s) : launch_rs<YMI_BF16, 2>(a, s);
}

}  // namespace ymi
