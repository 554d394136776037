// A whole C3 block with ONE Bottleneck and a 32-channel hidden width in ONE launch (gfx950) -- yolov5s' backbone.body.2
// (64 -> 64 at 160x160): cv3(cat(x1 + m.cv2(m.cv1(x1)), cv2(x))) with x1 = cv1(x).
//
// Replaces yolort/v5/models/common.py:172-173 (C3.forward) with :115-116 (Bottleneck.forward) inlined, each Conv being
// common.py:69-70 (SiLU(BN(conv))) with the BatchNorm folded on the host.
//
// Why: at 160x160 these five convolutions are HBM-streaming work.  Launched separately (cv1 + cv2 + m.cv1 in one launch,
// m.cv2, cv3: ops 2-4 of the yolov5s plan) they move 768 bytes per pixel -- 629 MB per 32-image batch, 196 us measured
// (profiles/r02z_layer_table_c2.csv) against 126 us at the 5 TB/s a streaming kernel reaches here; the block's input and
// output are 256 bytes per pixel (210 MB: 42 us).  Nothing between them has to exist in memory:
//   * a block of 8 waves owns a 16 x 16 output tile; every folded weight matrix of the C3 (cv1|cv2 8 KiB, m.cv1 2 KiB,
//     m.cv2 18 KiB, cv3 8 KiB) is RESIDENT in LDS in MFMA fragment order (loaded once per persistent block);
//   * phase A: over the 18 x 18 halo patch (11 pixel groups of 32), x goes global -> VGPR like the streaming 1x1 kernel,
//     x1 = cv1(x) and u = m.cv1(x1) are chained in registers (the rounded 16-byte output packets of a 1x1 ARE the
//     activation fragments of the next), u lands in an LDS patch in the resident-weights 3x3 kernel's slot layout
//     (zero outside the image: the 3x3's padding);
//   * phase B: for the wave's own 32 output pixels x1 and x2 = cv2(x) are computed once more in output order and stay in
//     registers (x1 is the Bottleneck's shortcut, x2 the second half of cv3's input);
//   * phase C: the 3x3 from the LDS patch (18 fragment reads, 18 MFMAs), SiLU, + x1, rounding;
//   * phase D: cv3 over [that | x2] from registers, SiLU, 16-byte NHWC stores.
// The halo costs 1.27x the x reads (L2 hits for the most part) and 17 % more MFMAs than the unfused form; every
// intermediate is rounded to the storage dtype exactly where the unfused launches round it and every accumulation runs in
// the same k order on top of the bias, so the result is BIT-IDENTICAL to the three-launch form (tests/test_c3_fused_gpu.py).
//
// Status (end of round 2): written without a GPU at hand -- cross-compiled for gfx950 and EXECUTED on the CPU simulator
// (tests/hipsim, tests/test_hipsim_kernels.py: bit-identical to the three launches), not yet timed.  Opt-in only
// (YOLORT_AMD_FUSE_C3=1), never taken by default; the A/B script is tools/gpu_calls/gpu_r3_c3fused.sh.
#include "conv_common.hpp"

namespace ymi {

struct C3Args {
    const uint16_t* x;
    uint16_t* y;
    const uint16_t *w12, *wm1, *wm2, *w3;
    const float *b12, *bm1, *bm2, *b3;
    int n, h, w, x_cs, y_cs;
    int k12, km1, km2, k3;   // row strides (k_pad) of the packed weight matrices
};

constexpr int F3_TH = 16, F3_TW = 16;             // output tile: 256 pixels, one 32-pixel group (2 rows) per wave
constexpr int F3_PW = F3_TW + 2, F3_PPIX = (F3_TH + 2) * (F3_TW + 2);   // halo patch 18 x 18 = 324 slots
constexpr int F3_HG = (F3_PPIX + 31) / 32;        // 11 halo pixel groups: waves 0-2 take two
// LDS map (bytes): weight fragments [frag index][64 lanes] x 16 B, biases [tile][4 groups][2 halves] x 16 B, patch [18 rows][18 slots x 80 B, padded to 1536 B]
constexpr int F3_W12 = 0;                         // (tile t in {cv1, cv2}, k16 step s): t*4 + s        8 KiB
constexpr int F3_WM1 = F3_W12 + 8 * 1024;         // (step s): s                                        2 KiB
constexpr int F3_WM2 = F3_WM1 + 2 * 1024;         // (tap, k16 half): tap*2 + ks                       18 KiB
constexpr int F3_W3 = F3_WM2 + 18 * 1024;         // (tile t, step s): t*4 + s                          8 KiB
constexpr int F3_BIAS = F3_W3 + 8 * 1024;         // tiles: 0, 1 = cv1, cv2; 2 = m.cv1; 3 = m.cv2; 4, 5 = cv3
constexpr int F3_PATCH = F3_BIAS + 6 * 128;
// patch slot = 64 B of channels + 16 B pad, patch row = 18 slots padded to 1536 B: with an 80-byte slot pitch 16 consecutive slots
// fall on 16 different 16-byte bank slots (5 c mod 16), and with a row pitch that is a multiple of 256 B the two half-rows a
// ds_read_b128 lane group spans (columns 0-3, 12-15 of one output row and 4-11 of the next) still cover 16 different ones: the
// fragment reads are conflict-free (4.0 LDS cycles per read in the bank model of MI355X_MICROARCH.md; 8.0 with a dense
// 18-slot row, which is also what the XOR swizzle of conv3x3_c32.hip gets: its measured 38 % conflict share).  Every tap's
// fragment address is the lane's base plus a CONSTANT (an instruction offset instead of nine address registers).  The
// column-wise ds_write_b128 of phase A cost 10.4 instead of 8 LDS cycles at row changes (2-4 writes per wave and tile).
constexpr int F3_SLOT = 80, F3_ROW = 1536;
constexpr int F3_LDS = F3_PATCH + (F3_TH + 2) * F3_ROW;   // 65 280 B: two blocks per CU, no opt-in needed (<= 64 KiB)

template <int DT>
__device__ __forceinline__ typename Mfma<DT>::frag as_frag(const u32x4& p) {
    typename Mfma<DT>::frag f;
    __builtin_memcpy(&f, &p, 16);
    return f;
}

__device__ __forceinline__ f32x16 bias_acc(const f32x4* bl, int tile, int hi) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = bl[(tile * 4 + g) * 2 + hi];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g * 4 + e] = b[e];
    }
    return acc;
}

// The lane swap of silu_pack_subtile is an involution: applied to a packet pair it returns the pre-swap words, which are
// exactly the layout a residual is loaded in (rv[g] = channels g*8 + hi*4 .. +3 of pixel lane & 31).
__device__ __forceinline__ void unswap_packets(const u32x4 (&o)[2], u32x2 (&rv)[4]) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
        const auto rx = __builtin_amdgcn_permlane32_swap(o[gq][0], o[gq][2], false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(o[gq][1], o[gq][3], false, false);
        const u32x2 r0 = {rx[0], ry[0]}, r1 = {rx[1], ry[1]};
        rv[2 * gq] = r0;
        rv[2 * gq + 1] = r1;
    }
}

template <int DT>
__global__ __launch_bounds__(512, 4) void c3_fused32_kernel(const C3Args a, int tiles_x, int tiles_y, int ntiles) {
    typedef typename Mfma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char f3_sm[];
    frag* const w12l = reinterpret_cast<frag*>(f3_sm + F3_W12);
    frag* const wm1l = reinterpret_cast<frag*>(f3_sm + F3_WM1);
    frag* const wm2l = reinterpret_cast<frag*>(f3_sm + F3_WM2);
    frag* const w3l = reinterpret_cast<frag*>(f3_sm + F3_W3);
    f32x4* const bl = reinterpret_cast<f32x4*>(f3_sm + F3_BIAS);
    unsigned char* const patch = f3_sm + F3_PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;

    // ---- resident weights: fragment (tile t, k16 step s) = rows t*32 + frow, k = 16*s + 8*hi .. +7 ----
    auto fill = [&](frag* dst, const uint16_t* w, int kstride, int nfrag, int ks) {
        for (int f = wave; f < nfrag; f += 8) {
            const int t = f / ks, s = f - t * ks;
            dst[f * 64 + lane] = *reinterpret_cast<const frag*>(w + (int64_t)(t * 32 + frow) * kstride + 16 * s + 8 * hi);
        }
    };
    fill(w12l, a.w12, a.k12, 8, 4);
    fill(wm1l, a.wm1, a.km1, 2, 2);
    fill(wm2l, a.wm2, a.km2, 18, 18);
    fill(w3l, a.w3, a.k3, 8, 4);
    if (tid < 48) {   // bias quad of (tile t, group g, half h): couts g*8 + h*4 .. of that tile
        const int t = tid >> 3, g = (tid >> 1) & 3, h = tid & 1;
        const float* src = t < 2 ? a.b12 + t * 32 : (t == 2 ? a.bm1 : (t == 3 ? a.bm2 : a.b3 + (t - 4) * 32));
        bl[tid] = *reinterpret_cast<const f32x4*>(src + g * 8 + h * 4);
    }

    // ---- phase C geometry (fixed per lane): output pixel p = wave*32 + frow -> (r, c); tap (dy, dx), k16 half ks reads
    //      patch + (r + dy)*1536 + (c + dx)*80 + ks*32 + hi*16 ----
    const int pr_o = (wave * 32 + frow) / F3_TW, pc_o = (wave * 32 + frow) % F3_TW;
    const unsigned char* const pc_base = patch + pr_o * F3_ROW + pc_o * F3_SLOT + hi * 16;
    const bool two = wave + 8 < F3_HG;   // wave-uniform: this wave has a second halo group (halo group hg = wave + 8*j covers patch slots hg*32 + frow)
    const u32x2 none[4] = {};
    __syncthreads();   // the resident weights are written

    for (int idx = blockIdx.x; idx < ntiles; idx += gridDim.x) {
        int t = xcd_remap(idx, ntiles);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        const int img = t / tiles_y;
        const int oy0 = ty * F3_TH, ox0 = tx * F3_TW;
        const uint16_t* const ximg = a.x + (int64_t)img * a.h * a.w * a.x_cs;

        // ---- global loads first: the centre group's x and the first halo group's x (clamped addresses; masked later) ----
        const int oy = oy0 + pr_o, ox = ox0 + pc_o;
        const bool okc = oy < a.h && ox < a.w;
        frag xb[4], xa[4];
        {
            const int cy = oy < a.h ? oy : a.h - 1, cx = ox < a.w ? ox : a.w - 1;
            const uint16_t* px = ximg + (int64_t)(cy * a.w + cx) * a.x_cs + 8 * hi;
#pragma unroll
            for (int s = 0; s < 4; ++s) xb[s] = *reinterpret_cast<const frag*>(px + 16 * s);
        }
        // halo group hg covers patch slots q = hg*32 + frow.  (The halo geometry is recomputed per tile on purpose: kept across
        // the 3x3 it costs registers this kernel does not have -- 128 per lane at four waves per SIMD.)
        int fr = frow;
        asm volatile("" : "+v"(fr));
        auto halo_load = [&](int hg, int& po, bool& inside) {   // po: byte offset of the lane's patch slot, -1 past the patch
            const int q = hg * 32 + fr;
            const int qc = q < F3_PPIX ? q : F3_PPIX - 1;
            const int pr = qc / F3_PW, pc = qc - pr * F3_PW;
            po = q < F3_PPIX ? pr * F3_ROW + pc * F3_SLOT + hi * 16 : -1;
            const int iy = oy0 - 1 + pr, ix = ox0 - 1 + pc;
            inside = q < F3_PPIX && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
            const int cy = iy < 0 ? 0 : (iy < a.h ? iy : a.h - 1), cx = ix < 0 ? 0 : (ix < a.w ? ix : a.w - 1);
            const uint16_t* px = ximg + (int64_t)(cy * a.w + cx) * a.x_cs + 8 * hi;
#pragma unroll
            for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const frag*>(px + 16 * s);
        };
        // phase A of one halo group: x1 = cv1(x), u = m.cv1(x1), chained in registers; zero outside the image (the 3x3's padding)
        auto halo_compute = [&](bool inside, u32x4 (&pu)[2]) {
            f32x16 acc = bias_acc(bl, 0, hi);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = Mfma<DT>::run(w12l[s * 64 + lane], xa[s], acc);
            u32x4 p1[2];
            silu_pack_subtile<DT, false, true>(acc, none, p1);
            f32x16 acc2 = bias_acc(bl, 2, hi);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc2 = Mfma<DT>::run(wm1l[s * 64 + lane], as_frag<DT>(p1[s]), acc2);
            silu_pack_subtile<DT, false, true>(acc2, none, pu);
            if (!inside) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                pu[0] = z;
                pu[1] = z;
            }
        };
        int q0, q1 = -1;
        bool in0, in1 = false;
        u32x4 pu0[2], pu1[2];
        halo_load(wave, q0, in0);
        halo_compute(in0, pu0);
        // (compiler fences between the phases: without them the weight fragments the phases share are read from LDS once and
        // carried through SCRATCH -- re-reading the resident image is the point of keeping it there)
        asm volatile("" ::: "memory");
        if (two) halo_load(wave + 8, q1, in1);   // into the registers the first group has just released; phase B covers the latency
        // ---- phase B: x1 and x2 = cv2(x) of the wave's own 32 output pixels, kept as packets ----
        u32x4 pk1[2], pk2[2];
        {
            f32x16 acc0 = bias_acc(bl, 0, hi), acc1 = bias_acc(bl, 1, hi);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = Mfma<DT>::run(w12l[s * 64 + lane], xb[s], acc0);
                acc1 = Mfma<DT>::run(w12l[(4 + s) * 64 + lane], xb[s], acc1);
            }
            silu_pack_subtile<DT, false, true>(acc0, none, pk1);
            silu_pack_subtile<DT, false, true>(acc1, none, pk2);
        }
        asm volatile("" ::: "memory");
        if (two) halo_compute(in1, pu1);
        __syncthreads();   // every wave is done reading the previous tile's patch
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) *reinterpret_cast<u32x4*>(patch + q0 + gq * 32) = pu0[gq];   // (first group: slots 0 .. 255, always inside the patch)
        if (two && q1 >= 0) {
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) *reinterpret_cast<u32x4*>(patch + q1 + gq * 32) = pu1[gq];
        }
        __syncthreads();   // the patch is complete

        // ---- phase C: m.cv2 (3x3) from the patch, SiLU, + x1 ----
        u32x4 pv[2];
        {
            f32x16 acc = bias_acc(bl, 3, hi);
#pragma unroll
            for (int ts = 0; ts < 18; ++ts) {   // (tap, k16 half)
                const frag fa = *reinterpret_cast<const frag*>(pc_base + ((ts >> 1) / 3) * F3_ROW + ((ts >> 1) % 3) * F3_SLOT + (ts & 1) * 32);
                acc = Mfma<DT>::run(wm2l[ts * 64 + lane], fa, acc);
            }
            u32x2 rv[4];
            unswap_packets(pk1, rv);
            silu_pack_subtile<DT, true, true>(acc, rv, pv);
        }
        // ---- phase D: cv3 over [that | x2], SiLU, stores ----
        {
            f32x16 acc0 = bias_acc(bl, 4, hi), acc1 = bias_acc(bl, 5, hi);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const frag xf = as_frag<DT>(s < 2 ? pv[s] : pk2[s - 2]);
                acc0 = Mfma<DT>::run(w3l[s * 64 + lane], xf, acc0);
                acc1 = Mfma<DT>::run(w3l[(4 + s) * 64 + lane], xf, acc1);
            }
            u32x4 o0[2], o1[2];
            silu_pack_subtile<DT, false, true>(acc0, none, o0);
            silu_pack_subtile<DT, false, true>(acc1, none, o1);
            if (okc) {
                uint16_t* yp = a.y + ((int64_t)(img * a.h + oy) * a.w + ox) * a.y_cs + hi * 8;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    *reinterpret_cast<u32x4*>(yp + q * 16) = o0[q];
                    *reinterpret_cast<u32x4*>(yp + 32 + q * 16) = o1[q];
                }
            }
        }
    }
}

template <int DT>
static int launch_c3_fused32(const C3Args& a, hipStream_t s) {
    const int tiles_x = cdiv(a.w, F3_TW), tiles_y = cdiv(a.h, F3_TH);
    const int ntiles = a.n * tiles_x * tiles_y;
    auto kfn = c3_fused32_kernel<DT>;
    static int per_cu = 0;   // persistent blocks: as many as are resident at once
    if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 512, F3_LDS) != hipSuccess || per_cu < 1)) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const int resident = per_cu * 256;   // MI355X: 256 CUs
    hipLaunchKernelGGL(kfn, dim3(ntiles < resident ? ntiles : resident), dim3(512), F3_LDS, s, a, tiles_x, tiles_y, ntiles);
    return check_launch("c3_fused32_kernel");
}

int c3_tile_launch(const ymi_c3_desc* d, hipStream_t s);   // c3_tile.hip: hidden widths 64 / 128, streamed weights

int c3_fused_launch(const ymi_c3_desc* d, hipStream_t s) {
    YMI_REQUIRE(d != nullptr, "ymi_c3_fused: null descriptor");
    if (d->c_hidden == 64 || d->c_hidden == 128) return c3_tile_launch(d, s);
    YMI_REQUIRE(d->mode == 0 && d->wblob == nullptr, "ymi_c3_fused: the resident-weights instance takes mode 0 without a weight stream");
    YMI_REQUIRE(d->x && d->y && d->w12 && d->b12 && d->wm1 && d->bm1 && d->wm2 && d->bm2 && d->w3 && d->b3, "ymi_c3_fused: null buffer");
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16, "ymi_c3_fused: 16-bit storage only");
    YMI_REQUIRE(d->c_in == 64 && d->c_hidden == 32 && d->c_out == 64 && d->n_bottlenecks == 1 && d->shortcut == 1,
                "ymi_c3_fused: this build holds the 64 -> 64 C3 with one shortcut Bottleneck of 32 hidden channels (got %d -> %d, hidden %d, n %d)", d->c_in,
                d->c_out, d->c_hidden, d->n_bottlenecks);
    YMI_REQUIRE(d->n >= 1 && d->h >= 1 && d->w >= 1 && d->x_cstride % 8 == 0 && d->x_cstride >= 64 && d->y_cstride % 8 == 0 && d->y_cstride >= 64,
                "ymi_c3_fused: views need 16-byte aligned pixels of at least 64 channels");
    YMI_REQUIRE(d->k12_pad >= 64 && d->km1_pad >= 32 && d->km2_pad >= 288 && d->k3_pad >= 64 && (d->k12_pad | d->km1_pad | d->km2_pad | d->k3_pad) % 8 == 0,
                "ymi_c3_fused: packed weight rows are too short (k_pad %d / %d / %d / %d)", d->k12_pad, d->km1_pad, d->km2_pad, d->k3_pad);
    YMI_REQUIRE((int64_t)d->h * d->w * (d->x_cstride > d->y_cstride ? d->x_cstride : d->y_cstride) < ((int64_t)1 << 31), "ymi_c3_fused: one image must stay below 2^31 elements");
    C3Args a;
    a.x = (const uint16_t*)d->x; a.y = (uint16_t*)d->y;
    a.w12 = (const uint16_t*)d->w12; a.wm1 = (const uint16_t*)d->wm1; a.wm2 = (const uint16_t*)d->wm2; a.w3 = (const uint16_t*)d->w3;
    a.b12 = d->b12; a.bm1 = d->bm1; a.bm2 = d->bm2; a.b3 = d->b3;
    a.n = d->n; a.h = d->h; a.w = d->w; a.x_cs = d->x_cstride; a.y_cs = d->y_cstride;
    a.k12 = d->k12_pad; a.km1 = d->km1_pad; a.km2 = d->km2_pad; a.k3 = d->k3_pad;
    return d->dtype == YMI_F16 ? launch_c3_fused32<YMI_F16>(a, s) : launch_c3_fused32<YMI_BF16>(a, s);
}

}  // namespace ymi

extern "C" int ymi_c3_fused(const ymi_c3_desc* d, void* stream) { return ymi::c3_fused_launch(d, (hipStream_t)stream); }
