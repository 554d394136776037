// Streaming 1x1 convolution for the memory-bound, shallow-K layers (gfx950): cin <= 128, every C3.cv1/cv2/cv3 and
// Bottleneck.cv1 at the 160x160 / 80x80 levels of yolov5s (320x320 / 160x160 of yolov5m).
//
// Those layers move 2-4 bytes per MAC; their bound is HBM (e.g. 64->64 at 160x160, batch 32: 210 MB = 26 us at 8 TB/s) and the
// tiled implicit-GEMM kernels reach a quarter of it: with K = 64 a 128-pixel block is two k-steps of work behind a full
// LDS-DMA prologue, one barrier pair and an epilogue, and too few bytes are in flight per CU.  Here nothing is staged and
// nothing synchronises:
//   * a WAVE owns 32-pixel groups (one MFMA column block) and ALL k of a 32*TNW-wide cout block; the folded weights (<= 32 KiB)
//     are loaded into LDS once per block, in fragment order, and re-read per group (one conflict-free 1 KiB sweep per MFMA);
//   * the activation fragments go global -> VGPR directly (lane l: 16 bytes = channels 8*(l>>5) + 16*s .. of pixel l & 31;
//     the k16 steps of a group together read each pixel's contiguous K*2 bytes), prefetched one group ahead, so every wave
//     keeps K/16 KiB in flight and a CU with 12 resident waves has >= 48 KiB outstanding -- what the HBM latency x
//     bandwidth product asks for (MI355X_MICROARCH.md: ~0.9k cycles, 31 B/clk/CU at 8 TB/s);
//   * MFMAs, bias + SiLU, the optional CHAINED 1x1 (Bottleneck.cv1 from the rounded outputs still in registers), the channel
//     split of a fused cv1 + cv2 pair and the 16-byte NHWC stores are the shared epilogue of conv_common.hpp.
// No activation staging, no barrier in the loop; persistent grid (waves stride over the groups).
//
// Replaces yolort/v5/models/common.py:69-70 (Conv.forward) / :172-173 (C3: cv1 and cv2 read the same input) for these shapes.
#include <cstdlib>

#include "conv_common.hpp"

namespace ymi {

// Lean epilogue of one 32-cout x 32-pixel sub-tile (cout tile fully inside cout: the launcher requires cout % 32 == 0): SiLU,
// optional residual, rounding, lanes l / l+32 exchange halves (v_permlane32_swap) so that every lane stores 8 consecutive
// channels = 16 bytes, channel split of a fused cv1 + cv2 pair.  Same arithmetic as finish_subtile (conv_common.hpp) -- that one
// batches loads and address math for whole wave tiles and needs the 256-register budget of the tiled kernels; this kernel wants
// 3-4 waves per SIMD.
struct StreamOut {
    uint16_t* y;
    uint16_t* y2;
    const uint16_t* res;
    int y_cs, y2_cs, split, res_cs, act;
};

// STORE = false: bias / SiLU / residual / rounding only, the packets come back through frag_out (the row-transposed store below)
template <int DT, bool FRAGS, bool STORE = true>
__device__ __forceinline__ void stream_store(const StreamOut& o, const f32x16& acc, int64_t m, bool ok, int cbase, int hi, u32x4* frag_out) {
    u32x2 rv[4] = {};
    const bool res = o.res != nullptr;   // wave-uniform
    if (res && ok) {
        const uint16_t* rp = o.res + m * o.res_cs + cbase + hi * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) rv[g] = *reinterpret_cast<const u32x2*>(rp + g * 8);
    }
    u32x4 pkt[2];   // packed math + hardware pair conversion (conv_common.hpp): lanes < 32 hold channels [g*8, g*8+8), lanes >= 32 [(g+1)*8, ..), g = 0, 2
    if (o.act == YMI_ACT_SILU) {
        if (res) silu_pack_subtile<DT, true, true>(acc, rv, pkt);
        else silu_pack_subtile<DT, false, true>(acc, rv, pkt);
    } else {
        if (res) silu_pack_subtile<DT, true, false>(acc, rv, pkt);
        else silu_pack_subtile<DT, false, false>(acc, rv, pkt);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if constexpr (FRAGS) frag_out[q] = pkt[q];   // the rounded 16-byte packet IS the activation fragment of the chained 1x1 (k16 step 2*tile + q)
        if (STORE && ok) {
            const int co = cbase + (2 * q + hi) * 8;
            uint16_t* yp;
            if (o.split > 0 && co >= o.split) yp = o.y2 + m * o.y2_cs + (co - o.split);
            else yp = o.y + m * o.y_cs + co;
            *reinterpret_cast<u32x4*>(yp) = pkt[q];
        }
    }
}

// Row-transposed store of a wave's 32-pixel x (32 * TNW)-cout block (tp_off >= 0): the epilogue packets (lanes l / l + 32 hold
// adjacent 16 bytes of pixel l & 31: one store instruction would write 32 bytes into each of 32 lines -- measured -11 % of the
// streaming bandwidth, tools/partial_line_bench.hip) go through a wave-private LDS tile [32 pixels][64 * TNW + 16 bytes] (the
// 16-byte pad makes the column-wise ds_write_b128 conflict-free) and leave as whole rows: 64 lanes x 16 bytes = 16 / TNW full rows
// per instruction.  No barrier: writer and reader are the same wave.
template <int TNW> constexpr int STREAM_TP_PITCH = 64 * TNW + 16;
template <int TNW> constexpr int STREAM_TP_BYTES = 32 * STREAM_TP_PITCH<TNW>;
template <int TNW> constexpr bool STREAM_TP_OK = (TNW == 1 || TNW == 2 || TNW == 4);

template <int DT, int TNW, int KS, bool CHAIN>
__global__ __launch_bounds__(256, 3) void conv1x1_stream_kernel(const ConvArgs a, int ngroups, int ncb, int tp_off) {
    typedef typename Mfma<DT>::frag frag;
    // The whole folded weight matrix (<= 128 x 128 x 2 B = 32 KiB) and the bias sit in LDS in FRAGMENT order: fragment
    // (cout tile t, k16 step s) is 64 lanes x 16 B contiguous, so a wave's read is one conflict-free 1 KiB sweep.  (Keeping the
    // fragments in registers instead costs 32-64 VGPRs that are live across the epilogue: measured 256 VGPRs + scratch.)
    extern __shared__ __attribute__((aligned(16))) unsigned char st_sm[];
    frag* wl = reinterpret_cast<frag*>(st_sm);                               // [cout_pad / 32][KS][64 lanes]
    const int ntile = a.cout_pad >> 5;
    f32x4* bl = reinterpret_cast<f32x4*>(st_sm + (size_t)ntile * KS * 1024);  // [cout_pad / 32][4 groups][2 halves]
    const int lane = threadIdx.x & 63;
    const int hi = lane >> 5, frow = lane & 31;
    for (int f = threadIdx.x >> 6; f < ntile * KS; f += 4) {
        const int t = f / KS, s = f - t * KS;
        wl[f * 64 + lane] = *reinterpret_cast<const frag*>(a.w + (int64_t)(t * 32 + frow) * a.k_pad + 16 * s + 8 * hi);
    }
    for (int i = threadIdx.x; i < ntile * 8; i += 256) {   // bias quad of (tile t, group g, half h): couts t*32 + g*8 + h*4 ..
        const int t = i >> 3, g = (i >> 1) & 3, h = i & 1;
        bl[i] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + g * 8 + h * 4);
    }
    // chained 1x1 (ymi_conv_desc.chain_w): its folded weights [chain_cout][chain_k = 32*TNW] and bias, same fragment order
    constexpr int KS2 = 2 * TNW;
    const int ntile2 = CHAIN ? a.chain_cout >> 5 : 0;
    frag* wl2 = reinterpret_cast<frag*>(st_sm + (size_t)ntile * KS * 1024 + (size_t)ntile * 8 * 16);
    f32x4* bl2 = reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(wl2) + (size_t)ntile2 * KS2 * 1024);
    if constexpr (CHAIN) {
        for (int f = threadIdx.x >> 6; f < ntile2 * KS2; f += 4) {
            const int t = f / KS2, s2 = f - t * KS2;
            wl2[f * 64 + lane] = *reinterpret_cast<const frag*>(a.chain_w + (int64_t)(t * 32 + frow) * (32 * TNW) + 16 * s2 + 8 * hi);
        }
        for (int i = threadIdx.x; i < ntile2 * 8; i += 256) {
            const int t = i >> 3, g = (i >> 1) & 3, h = i & 1;
            bl2[i] = *reinterpret_cast<const f32x4*>(a.chain_bias + t * 32 + g * 8 + h * 4);
        }
    }
    __syncthreads();
    StreamOut o1, o2;
    o1.y = reinterpret_cast<uint16_t*>(a.y); o1.y2 = reinterpret_cast<uint16_t*>(a.y2); o1.res = a.res;
    o1.y_cs = a.y_cs; o1.y2_cs = a.y2_cs; o1.split = a.split; o1.res_cs = a.res_cs; o1.act = a.act;
    o2.y = reinterpret_cast<uint16_t*>(a.chain_y); o2.y2 = nullptr; o2.res = nullptr;
    o2.y_cs = a.chain_y_cs; o2.y2_cs = 0; o2.split = 0; o2.res_cs = 0; o2.act = YMI_ACT_SILU;

    const int wave_g = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * 4;                     // a multiple of ncb: a wave keeps its cout block for every group it visits
    const int cb = wave_g % ncb;
    const int c0 = cb * (32 * TNW);
    const int t0 = cb * TNW;
    const int first = wave_g / ncb, stride = nwaves / ncb;
    auto load_group = [&](int g, frag (&xf)[KS]) {
        int m = g * 32 + frow;
        m = m < a.M ? m : a.M - 1;                        // clamped: lanes past M read a valid pixel and are masked at the store
        const uint16_t* px = a.x + (int64_t)m * a.x_cs + 8 * hi;
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const frag*>(px + 16 * s);
    };
    frag x0[KS], x1[KS];                                  // two-deep register ring; >= 3 waves per SIMD hide the rest of the latency
    if (first < ngroups) load_group(first, x0);

    for (int g = first; g < ngroups; g += stride) {
        if (g + stride < ngroups) load_group(g + stride, x1);
        f32x16 acc[TNW][1];
#pragma unroll
        for (int i = 0; i < TNW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = t0 + i < ntile ? t0 + i : ntile - 1;   // a partial last block re-uses the last tile (never stored)
                const f32x4 b = bl[(tt * 4 + q) * 2 + hi];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][0][q * 4 + e] = b[e];
            }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < TNW; ++i) {
                const int tt = t0 + i < ntile ? t0 + i : ntile - 1;
                acc[i][0] = Mfma<DT>::run(wl[(tt * KS + s) * 64 + lane], x0[s], acc[i][0]);
            }
        const int64_t m = (int64_t)g * 32 + frow;
        const bool ok = m < a.M;
        bool chained = false;
        if constexpr (CHAIN) chained = c0 == 0;            // the first cout block is the chained conv's input (wave-uniform)
        if (chained) {
            u32x4 fr[TNW][2];
#pragma unroll
            for (int i = 0; i < TNW; ++i) stream_store<DT, true>(o1, acc[i][0], m, ok, i * 32, hi, fr[i]);
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                if (i2 < ntile2) {
                    f32x16 acc2;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 b = bl2[(i2 * 4 + q) * 2 + hi];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[q * 4 + e] = b[e];
                    }
#pragma unroll
                    for (int s2 = 0; s2 < KS2; ++s2) {
                        frag xf;
                        __builtin_memcpy(&xf, &fr[s2 >> 1][s2 & 1], 16);
                        acc2 = Mfma<DT>::run(wl2[(i2 * KS2 + s2) * 64 + lane], xf, acc2);
                    }
                    stream_store<DT, false>(o2, acc2, m, ok, i2 * 32, hi, nullptr);
                }
            }
        } else if (STREAM_TP_OK<TNW> && tp_off >= 0 && c0 < a.cout) {   // (blocks past cout exist when cout_pad > cout: never stored, below)
            constexpr int PITCH = STREAM_TP_PITCH<TNW>, LPR = 4 * TNW, RPI = 64 / LPR;   // lanes per row, rows per store instruction
            unsigned char* tw = st_sm + tp_off + (threadIdx.x >> 6) * STREAM_TP_BYTES<TNW>;
#pragma unroll
            for (int i = 0; i < TNW; ++i) {
                u32x4 pk[2];
                stream_store<DT, true, false>(o1, acc[i][0], m, ok, c0 + i * 32, hi, pk);
#pragma unroll
                for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(tw + frow * PITCH + (i * 4 + 2 * q + hi) * 16) = pk[q];
            }
            __builtin_amdgcn_wave_barrier();   // (scheduling fence only: the LDS executes a wave's operations in order)
            // the wave's whole cout block has ONE destination (the launcher requires split % block width == 0)
            const bool second = o1.split > 0 && c0 >= o1.split;
            uint16_t* yb = second ? o1.y2 + (c0 - o1.split) : o1.y + c0;
            const int ycs = second ? o1.y2_cs : o1.y_cs;
            const int row_l = lane / LPR, chunk = lane - row_l * LPR;
#pragma unroll
            for (int j = 0; j < 2 * TNW; ++j) {
                const int row = j * RPI + row_l;
                const u32x4 v = *reinterpret_cast<const u32x4*>(tw + row * PITCH + chunk * 16);
                const int64_t mr = (int64_t)g * 32 + row;
                if (mr < a.M) *reinterpret_cast<u32x4*>(yb + mr * ycs + chunk * 8) = v;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int i = 0; i < TNW; ++i)
                if (c0 + i * 32 < a.cout) stream_store<DT, false>(o1, acc[i][0], m, ok, c0 + i * 32, hi, nullptr);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) x0[s] = x1[s];       // rotate the ring (register moves; the loads stay in flight)
    }
}

template <int DT, int TNW, int KS>
static int launch_stream(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int ngroups = cdiv(a.M, 32);
    const int ncb = cdiv(a.cout_pad, 32 * TNW);
    size_t lds = (size_t)(a.cout_pad / 32) * KS * 1024 + (size_t)(a.cout_pad / 32) * 8 * 16;
    if (a.chain_w != nullptr) lds += (size_t)(a.chain_cout / 32) * (2 * TNW) * 1024 + (size_t)(a.chain_cout / 32) * 8 * 16;
    // persistent grid = the blocks that are RESIDENT at once (register count and LDS decide: 3 or 4 per CU for these
    // instantiations), a multiple of ncb.  A fixed 4 blocks per CU left the 158-VGPR instances (3 blocks per CU) with a
    // fourth, non-resident quarter of the grid that started only when the first blocks had finished their whole share.
    // row-transposed stores: whole cout blocks only (a ragged last block keeps the per-packet stores)
    static const char* tp_env = getenv("YOLORT_AMD_STREAM_TP");   // A/B: "0" = per-packet stores everywhere
    int tp_off = -1;
    // (and only while the block stays inside the 64 KiB of dynamic LDS a launch gets without the opt-in: 128 -> 128 with
    // 128-wide blocks would not)
    if (STREAM_TP_OK<TNW> && a.cout % (32 * TNW) == 0 && lds + (size_t)4 * STREAM_TP_BYTES<TNW> <= 64 * 1024 && !(tp_env && tp_env[0] == '0')) {
        tp_off = (int)lds;
        lds += (size_t)4 * STREAM_TP_BYTES<TNW>;
    }
    const bool chain = a.chain_w != nullptr;
    static size_t occ_lds[2] = {0, 0};
    static int occ_blocks[2] = {0, 0};
    if (occ_blocks[chain] == 0 || occ_lds[chain] != lds) {
        int per_cu = 0;
        const hipError_t e = chain ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1x1_stream_kernel<DT, TNW, KS, true>, 256, lds)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1x1_stream_kernel<DT, TNW, KS, false>, 256, lds);
        if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; }
        occ_blocks[chain] = per_cu > 4 ? 4 : per_cu;   // 16 waves per CU already cover the latency x bandwidth product
        occ_lds[chain] = lds;
    }
    static const char* fixed_env = getenv("YOLORT_AMD_STREAM_BLOCKS");   // tuning aid: blocks per CU of the persistent grid (A/B against the occupancy-sized grid)
    int blocks = 256 * (fixed_env ? atoi(fixed_env) : occ_blocks[chain]);
    const int need = cdiv(ngroups * ncb, 4);
    if (blocks > need) blocks = need;
    blocks = cdiv(blocks * 4, ncb * 4) * ncb;             // waves = 4 * blocks: multiple of ncb
    if (blocks < 1) blocks = ncb;
    if (a.chain_w != nullptr) hipLaunchKernelGGL((conv1x1_stream_kernel<DT, TNW, KS, true>), dim3(blocks), dim3(256), lds, s, a, ngroups, ncb, tp_off);
    else hipLaunchKernelGGL((conv1x1_stream_kernel<DT, TNW, KS, false>), dim3(blocks), dim3(256), lds, s, a, ngroups, ncb, tp_off);
    return check_launch("conv1x1_stream_kernel");
}

template <int DT, int TNW>
static int stream_ks(const ConvArgs& a, hipStream_t s) {
    switch (a.cin / 16) {
        case 2: return launch_stream<DT, TNW, 2>(a, s);
        case 4: return launch_stream<DT, TNW, 4>(a, s);
        case 6: return launch_stream<DT, TNW, 6>(a, s);
        case 8: return launch_stream<DT, TNW, 8>(a, s);
        default: break;
    }
    set_error("ymi_conv2d: the streaming 1x1 kernel has no instance for cin %d with %d-wide cout blocks", a.cin, 32 * TNW);
    return YMI_EINVAL;
}

// variant = TNW (cout block of 32 * TNW channels per wave); with a split / chained conv the block width must equal the split
int conv1x1_stream_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(a.kh == 1 && a.kw == 1 && a.sh == 1 && a.sw == 1 && a.ph == 0 && a.pw == 0 && a.cin % 32 == 0 && a.cin <= 128 && a.k_pad == a.cin,
                "ymi_conv2d: the streaming kernel handles 1x1 stride-1 convolutions with cin in {32, 64, 96, 128}");
    YMI_REQUIRE(out_dtype == dtype, "ymi_conv2d: the streaming 1x1 kernel stores the compute dtype");
    YMI_REQUIRE(a.x_cs % 8 == 0 && a.up2 == 0, "ymi_conv2d: the streaming 1x1 kernel needs x_cstride %% 8 == 0 and has no upsampled second output");
    YMI_REQUIRE(variant >= 1 && variant <= 4, "ymi_conv2d: streaming 1x1 variant (cout tiles per wave) must be 1..4");
    const int bw = 32 * variant;
    YMI_REQUIRE(a.split == 0 || a.split % bw == 0, "ymi_conv2d: the cout block width %d must divide the channel split %d", bw, a.split);
    YMI_REQUIRE(a.chain_w == nullptr || (a.chain_k == bw && a.chain_x2 == nullptr), "ymi_conv2d: a chained 1x1 needs %d-wide cout blocks here (and no second source)", a.chain_k);
    YMI_REQUIRE(a.cout % 32 == 0 && a.y_cs % 8 == 0 && (a.split == 0 || a.y2_cs % 8 == 0) && (a.res == nullptr || a.res_cs % 4 == 0),
                "ymi_conv2d: the streaming 1x1 kernel needs cout %% 32 == 0 and 16-byte aligned output rows");
#define YMI_STREAM(DT_)                                               \
    switch (variant) {                                                \
        case 1: return stream_ks<DT_, 1>(a, s);                      \
        case 2: return stream_ks<DT_, 2>(a, s);                      \
        case 3: return stream_ks<DT_, 3>(a, s);                      \
        default: return stream_ks<DT_, 4>(a, s);                     \
    }
    if (dtype == YMI_F16) { YMI_STREAM(YMI_F16) }
    YMI_STREAM(YMI_BF16)
#undef YMI_STREAM
}

}  // namespace ymi
