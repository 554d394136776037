// The first two layers of the r6.0 backbone in ONE launch (gfx950), fed from the planar input images (fixed-size streams) or from
// the letterboxed NHWC4 canvas (dynamic-shape streams):
//     body.0  Conv(3, 32, k=6, s=2, p=2) + BN + SiLU      (reference yolort/models/darknetv6.py:81, v5/models/common.py:69-70)
//     body.1  Conv(32, 64, k=3, s=2, p=1) + BN + SiLU     (darknetv6.py:85-86)
//
// Separately (conv_stem_planar_kernel, conv3x3_c32_kernel<S = 2>) the stem's output -- the largest activation of the network,
// 32 channels at H/2 x W/2: 210 MB per 32-image batch at 640 x 640 -- is written to HBM and read straight back: 420 MB of the
// step's 3.3 GB, and both kernels wait on memory, not on the matrix cores (90 + 75 us against 39 + 39 us of HBM time).  Here a
// persistent 8-wave block owns 8 x 16 output pixels of body.1 per tile:
//   stage 1  the 17 x 33 stem pixels the tile needs (18 groups of 32; 10 % halo overhead) are computed from the planar patch in LDS
//            exactly like conv_stem_planar_kernel computes them (same fragments, same K order on top of the bias, same SiLU and
//            rounding) and written -- rounded to the storage type, zero outside the stem's output (body.1's padding) -- into the
//            LDS patch conv3x3_c32_kernel<2> reads: columns split by parity, 64-byte pixels, 16-byte chunks XOR-swizzled;
//   stage 2  the 3x3 stride-2 convolution from that patch, folded weights resident in LDS in fragment order, one 32-pixel x
//            32-cout tile per wave, the shared epilogue.
// Both weight matrices are loaded once per block (the stem's 9 KiB into registers, body.1's 36 KiB into LDS); the planar patch of tile i+1 is DMA'd while stage 2 of tile i runs.
// Results are BIT-IDENTICAL to the two separate launches (tests/test_hipsim_kernels.py on the CPU simulator, tests/test_ops_gpu.py).
#include "conv_common.hpp"

namespace ymi {

constexpr int SB_TH = 8, SB_TW = 16;                     // body.1 output tile
constexpr int SB_PH = 2 * SB_TH + 1, SB_PW = 2 * SB_TW + 1;   // stem pixels per tile: 17 x 33
constexpr int SB_NE = SB_TW + 1;                         // even columns first (17), then the 16 odd ones (conv3x3_c32_kernel<2>'s patch order)
constexpr int SB_PPIX = SB_PH * SB_PW;                   // 561
constexpr int SB_GROUPS = (SB_PPIX + 31) / 32;           // 18 groups of 32 stem pixels
constexpr int SB_IR = 2 * SB_PH + 4, SB_IC = 80;         // planar patch: 38 rows x 80 pixels (needed: columns 4 .. 73) per plane
constexpr int SB_ENTRIES = 3 * SB_IR * (SB_IC / 8);      // 1140 16-byte row segments
constexpr int SB_PIECES = (SB_ENTRIES + 63) / 64;        // 18 DMA pieces of 1 KiB
// NHWC4 form (the letterboxed canvas of a dynamic-shape stream, 8 halves per super-pixel = 2 pixels x RGB0): 38 rows x 35 super-pixels per tile
constexpr int SB_SW = SB_PW + 2;                         // 35 super-pixel columns: stem pixel sx reads super-pixels sx - 1 .. sx + 1
constexpr int SB_SENTRIES = SB_IR * SB_SW;               // 1330 16-byte super-pixels
constexpr int SB_SPIECES = (SB_SENTRIES + 63) / 64;      // 21 DMA pieces of 1 KiB
constexpr int SB_MAX_IMGS = 32;
#ifndef YMI_SB_PK
#define YMI_SB_PK 0
#endif
constexpr bool SB_PK = YMI_SB_PK != 0;   // stage 1 SiLU with packed fp32 instructions (1) or scalar ones (0, default: 125 -> 122 us same-box, profiles/r03n)

struct SbImgs {
    const uint16_t* img[SB_MAX_IMGS];
};

constexpr int SB_W1 = 0, SB_W2 = 36 * 1024, SB_B2 = 256, SB_PLANAR = SB_SPIECES * 1024, SB_PATCH = ((SB_PPIX + 15) / 16) * 1024;   // (input patch sized for the larger, NHWC4 form)
constexpr int SB_LDS = SB_W1 + SB_W2 + SB_B2 + SB_PLANAR + SB_PATCH;   // 36 + 0.25 + 21 + 36 KiB (the stem weights live in registers)

// PLANAR: the input is one (3, H, W) planar image per batch element (pl.img; fixed-size streams); else a1.x is the NHWC4 canvas (n, H, W/2, 8) the letterbox wrote
template <int DT, bool PLANAR>
__global__ __launch_bounds__(512, 1) void stem_body1_fused_kernel(const ConvArgs a1, const ConvArgs a2, const SbImgs pl, int tiles_x, int tiles_y, int ntiles) {
    typedef typename Mfma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_sm[];
    frag* w2l = reinterpret_cast<frag*>(sb_sm + SB_W1);                                 // body.1 weights [(tap*2 + ks)*2 + i][64 lanes] x 16 B
    f32x4* b2l = reinterpret_cast<f32x4*>(sb_sm + SB_W1 + SB_W2);                       // body.1 bias    [2 tiles][4 groups][2 halves]
    uint16_t* planar = reinterpret_cast<uint16_t*>(sb_sm + SB_W1 + SB_W2 + SB_B2);      // [3 planes][38 rows][80 px]
    unsigned char* patch = sb_sm + SB_W1 + SB_W2 + SB_B2 + SB_PLANAR;                   // conv3x3_c32_kernel<2>'s patch: slot q = row * 33 + parity-split column

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, px = lane & 31;
    const int H = a1.h, W = 2 * a1.w_in;          // input image size in pixels (the stem's geometry counts super-pixels)

    // ---- resident weights ----
    {
        for (int f = wave; f < 36; f += 8) {                                 // body.1: fragment (ts, i) = rows i*32 + px, k = ts*16 + 8 hi .. +7
            const int i = f & 1, ts = f >> 1;
            w2l[f * 64 + lane] = *reinterpret_cast<const frag*>(a2.w + (int64_t)(i * 32 + px) * a2.k_pad + ts * 16 + hi * 8);
        }
        if (tid < 16) {
            const int t = tid >> 3, g = (tid >> 1) & 3, h = tid & 1;
            b2l[tid] = *reinterpret_cast<const f32x4*>(a2.bias + t * 32 + g * 8 + h * 4);
        }
    }
    f32x4 bias1[1][4];
    load_bias<1>(a1, 0, hi, bias1);
    // stem weights: the 9 k16 fragments of this lane's cout row stay in REGISTERS for the whole kernel (36 VGPRs; one 8-wave block per CU leaves 256 per lane)
    frag wf1[9];
    {
        const uint16_t* wr = a1.w + (int64_t)px * a1.k_pad + 8 * hi;       // fragment s = rows 0..31, k = 16 s + 8 hi .. +7
#pragma unroll
        for (int s = 0; s < 9; ++s) wf1[s] = *reinterpret_cast<const frag*>(wr + 16 * s);
    }

    // ---- input patch DMA geometry (fixed per lane).  PLANAR: entry e = (plane, row, segment of 8 pixels); NHWC4: entry e = (row, super-pixel column) ----
    int e_plane[3], e_row[3], e_col[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = (wave * 3 + j) * 64 + lane;
        if constexpr (PLANAR) {
            const int ec = e < SB_ENTRIES ? e : SB_ENTRIES - 1;
            const int plane = ec / (SB_IR * (SB_IC / 8));
            const int rem = ec - plane * (SB_IR * (SB_IC / 8));
            const int pr = rem / (SB_IC / 8), seg = rem - pr * (SB_IC / 8);
            e_plane[j] = e < SB_ENTRIES ? plane : -1;
            e_row[j] = pr - 4;
            e_col[j] = 8 * seg - 8;
        } else {
            const int ec = e < SB_SENTRIES ? e : SB_SENTRIES - 1;
            const int pr = ec / SB_SW, sc = ec - pr * SB_SW;
            e_plane[j] = e < SB_SENTRIES ? 0 : -1;
            e_row[j] = pr - 4;
            e_col[j] = sc - 2;     // super-pixel column relative to 2 * ox0
        }
    }
    auto tile_origin = [&](int idx, int& img, int& oy0, int& ox0) {
        int t = xcd_remap(idx, ntiles);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        img = t / tiles_y;
        oy0 = ty * SB_TH;
        ox0 = tx * SB_TW;
    };
    auto issue_planar = [&](int idx) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        if constexpr (PLANAR) {
            const uint16_t* base = pl.img[img];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pi = wave * 3 + j;
                if (pi < SB_PIECES) {   // wave-uniform
                    const int iy = 4 * oy0 + e_row[j], ix = 4 * ox0 + e_col[j];
                    const bool ok = (e_plane[j] >= 0) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);   // W % 8 == 0: a segment is in or out as a whole
                    const uint16_t* src = ok ? base + ((int64_t)e_plane[j] * H + iy) * W + ix : a1.zeros;
                    glds16(src, planar + pi * 512);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pi = wave * 3 + j;
                if (pi < SB_SPIECES) {   // wave-uniform
                    const int iy = 4 * oy0 + e_row[j], isp = 2 * ox0 + e_col[j];
                    const bool ok = (e_plane[j] >= 0) && ((unsigned)iy < (unsigned)H) && ((unsigned)isp < (unsigned)a1.w_in);
                    const uint16_t* src = ok ? a1.x + ((int64_t)(img * H + iy) * a1.w_in + isp) * a1.x_cs : a1.zeros;
                    glds16(src, planar + pi * 512);
                }
            }
        }
    };

    // ---- stage-1 geometry (fixed per lane): group g -> stem pixel (pr, pc) of the tile's 17 x 33 patch, its slot in the patch ----
    int s1_off[3], s1_rc[3], s1_dst[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int g = wave + 8 * j;
        const int q = g * 32 + px;
        const int qc = q < SB_PPIX ? q : SB_PPIX - 1;
        const int pr = qc / SB_PW, rem = qc - pr * SB_PW;
        const int pc = rem < SB_NE ? 2 * rem : 2 * (rem - SB_NE) + 1;
        s1_rc[j] = q < SB_PPIX ? ((pr << 16) | pc) : -1;
        s1_off[j] = PLANAR ? (2 * pr) * SB_IC + 2 * pc + 4              // planar patch element of tap (ky, kx') = (0, 0)
                           : ((2 * pr) * SB_SW + pc) * 8;                // NHWC4: super-pixel (row 2 pr, column pc) in halves
        s1_dst[j] = qc * 64 + ((hi ^ ((qc >> 2) & 3)) * 16);            // byte address of chunk `hi` of the slot (chunk hi + 2: ^ 32)
    }
    // ---- stage-2 geometry (fixed per lane): wave -> (32-pixel group, cout tile) ----
    const int pg = wave & 3, ct = wave >> 2;
    const int pr_o = (pg * 32 + px) / SB_TW, pc_o = (pg * 32 + px) % SB_TW;
    int ea[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        const int col = 2 * pc_o + dx;
        const int slot = (col & 1) ? SB_NE + (col >> 1) : (col >> 1);
        const int q = (2 * pr_o + dy) * SB_PW + slot;
        ea[t] = (q * 32 + ((hi ^ ((q >> 2) & 3)) * 8)) * 2;
    }
    constexpr int PLANE_HALFS = SB_IR * SB_IC;

    int idx = blockIdx.x;
    if (idx < ntiles) issue_planar(idx);
    // (A counted wait -- vmcnt(2): only the previous tile's two output stores may stay outstanding -- measured equal to waiting for everything:
    //  profiles/r03g_fused_stem_wait_ab.txt.  What bounds the tile is the vector ALU: 32 quarter-rate transcendentals (v_exp_f32 + v_rcp_f32 of the
    //  SiLUs) and ~95 other VALU instructions per 32-pixel stem group against 9 MFMAs.)
    for (; idx < ntiles; idx += gridDim.x) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the planar patch has landed; everyone is done reading the previous tile's stem patch (first pass: the weights are written)

        // ---- stage 1: stem pixels of the tile -> LDS patch ----
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (wave + 8 * j < SB_GROUPS) {   // wave-uniform
                f32x16 acc[1][1];
                init_acc<1, 1>(acc, bias1);
                if constexpr (PLANAR) {
                    // all 27 planar reads of the group first, then the 9 MFMAs: left to itself the compiler reads, waits and multiplies step by step
                    // (one LDS round trip per MFMA, two waves per SIMD to hide it -- measured 5.4 us per tile)
                    uint32_t rr[9], gg[9], bb[9];
#pragma unroll
                    for (int s = 0; s < 9; ++s) {
                        // tap = 2s + hi -> (ky, kx') = (tap / 3, tap % 3): compile-time per half
                        const int tap0 = 2 * s, tap1 = 2 * s + 1;
                        const int o0 = (tap0 / 3) * SB_IC + 2 * (tap0 % 3), o1 = (tap1 / 3) * SB_IC + 2 * (tap1 % 3);
                        const uint16_t* p0 = planar + s1_off[j] + (hi ? o1 : o0);
                        rr[s] = *reinterpret_cast<const uint32_t*>(p0);
                        gg[s] = *reinterpret_cast<const uint32_t*>(p0 + PLANE_HALFS);
                        bb[s] = *reinterpret_cast<const uint32_t*>(p0 + 2 * PLANE_HALFS);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 9; ++s) {
                        const uint32_t r = rr[s], g = gg[s], b = bb[s];
                        u32x4 q;
                        q[0] = (r & 0xffffu) | (g << 16);        // R0 G0
                        q[1] = b & 0xffffu;                      // B0 0
                        q[2] = (r >> 16) | (g & 0xffff0000u);    // R1 G1
                        q[3] = b >> 16;                          // B1 0
                        frag af;
                        __builtin_memcpy(&af, &q, 16);
                        acc[0][0] = Mfma<DT>::run(wf1[s], af, acc[0][0]);
                    }
                } else {
                    // NHWC4 canvas: a super-pixel IS the fragment of its tap (conv_stem_kernel's reads): nine 16-byte reads, then the nine MFMAs
                    frag af[9];
#pragma unroll
                    for (int s = 0; s < 9; ++s) {
                        const int tap0 = 2 * s, tap1 = 2 * s + 1;
                        const int o0 = ((tap0 / 3) * SB_SW + (tap0 % 3)) * 8, o1 = ((tap1 / 3) * SB_SW + (tap1 % 3)) * 8;
                        af[s] = *reinterpret_cast<const frag*>(planar + s1_off[j] + (hi ? o1 : o0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 9; ++s) acc[0][0] = Mfma<DT>::run(wf1[s], af[s], acc[0][0]);
                }
                const u32x2 norv[4] = {};
                u32x4 o[2];
                silu_pack_subtile<DT, false, true, SB_PK>(acc[0][0], norv, o);   // the epilogue arithmetic of the separate launch: SiLU, rounding, lane swap
                // body.1 pads with zeros: stem pixels outside the stem's output are 0, not SiLU(bias)
                const int sy = 2 * oy0 - 1 + (s1_rc[j] >> 16), sx = 2 * ox0 - 1 + (s1_rc[j] & 0xffff);
                const bool in = s1_rc[j] >= 0 && ((unsigned)sy < (unsigned)a1.ho) && ((unsigned)sx < (unsigned)a1.wo);
                if (!in) {
                    o[0] = u32x4{0u, 0u, 0u, 0u};
                    o[1] = u32x4{0u, 0u, 0u, 0u};
                }
                if (s1_rc[j] >= 0) {   // o[q] = channels [(2q + hi) * 8, +8): chunk 2q + hi of the slot
                    *reinterpret_cast<u32x4*>(patch + s1_dst[j]) = o[0];
                    *reinterpret_cast<u32x4*>(patch + (s1_dst[j] ^ 32)) = o[1];
                }
            }
        }
        __syncthreads();   // the stem patch is complete; the planar patch is free
        if (idx + (int)gridDim.x < ntiles) issue_planar(idx + gridDim.x);

        // ---- stage 2: 3x3 stride-2 convolution from the patch (conv3x3_c32_kernel<DT, 2, 2>, one cout tile per wave) ----
        f32x16 acc2[1][1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = b2l[(ct * 4 + g) * 2 + hi];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2[0][0][g * 4 + e] = b[e];
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // 18 (tap, k16 half) steps in two batches of nine: the 18 LDS reads of a batch first, then its MFMAs
            frag fa[9], fw[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int ts = half * 9 + u;
                fa[u] = *reinterpret_cast<const frag*>(patch + ((ts & 1) ? (ea[ts >> 1] ^ 32) : ea[ts >> 1]));
                fw[u] = w2l[(ts * 2 + ct) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 9; ++u) acc2[0][0] = Mfma<DT>::run(fw[u], fa[u], acc2[0][0]);
        }
        finish_wave_tile<DT, DT, 1, 1>(a2, acc2, ct * 32, hi, [&](int, int64_t& m, bool& ok) {
            const int oy = oy0 + pr_o, ox = ox0 + pc_o;
            ok = oy < a2.ho && ox < a2.wo;
            m = ((int64_t)img * a2.ho + oy) * a2.wo + ox;
        });
    }
}

template <int DT, bool PLANAR>
static int stem_body1_launch_t(const ConvArgs& s0, const ConvArgs& b0, const void* const* imgs, hipStream_t st) {
    const int tiles_x = cdiv(b0.wo, SB_TW), tiles_y = cdiv(b0.ho, SB_TH);
    auto kfn = stem_body1_fused_kernel<DT, PLANAR>;
    { const int rc_lds = allow_big_lds((const void*)kfn, SB_LDS); if (rc_lds != YMI_OK) return rc_lds; }
    const int per_launch = PLANAR ? SB_MAX_IMGS : b0.n;   // planar image pointers travel as kernel arguments, 32 at a time; the canvas is one tensor
    for (int base = 0; base < b0.n; base += per_launch) {
        ConvArgs a1 = s0, a2 = b0;
        const int n = b0.n - base < per_launch ? b0.n - base : per_launch;
        a1.n = a2.n = n;
        SbImgs pl;
        for (int i = 0; i < SB_MAX_IMGS; ++i) pl.img[i] = PLANAR ? (const uint16_t*)imgs[base + (i < n ? i : 0)] : nullptr;
        a2.y = (void*)((uint16_t*)b0.y + (int64_t)base * b0.ho * b0.wo * b0.y_cs);
        a2.M = n * a2.ho * a2.wo;
        const int ntiles = n * tiles_x * tiles_y;
        a2.nblk_m = ntiles;
        a2.nblk_n = 1;
        const int resident = 256;   // one 8-wave block per CU (102 KiB of LDS)
        hipLaunchKernelGGL(kfn, dim3(ntiles < resident ? ntiles : resident), dim3(512), SB_LDS, st, a1, a2, pl, tiles_x, tiles_y, ntiles);
    }
    return check_launch("stem_body1_fused_kernel");
}

static int stem_body1_check(const ConvArgs& a1, const ConvArgs& a2, const char* who) {
    YMI_REQUIRE(a1.cin == 8 && a1.kh == 6 && a1.kw == 3 && a1.sh == 2 && a1.sw == 1 && a1.ph == 2 && a1.pw == 1 && a1.k_pad >= 144 && a1.cout == 32 && a1.cout_pad >= 32 &&
                    a1.zeros != nullptr && a1.split == 0 && a1.res == nullptr && a1.act == YMI_ACT_SILU && a1.chain_w == nullptr && a1.up2 == 0,
                "%s: the first descriptor must be the 32-channel stem in its 6x3 s(2,1) p(2,1) super-pixel form (SiLU, no residual / split / chain)", who);
    YMI_REQUIRE(a2.kh == 3 && a2.kw == 3 && a2.sh == 2 && a2.sw == 2 && a2.ph == 1 && a2.pw == 1 && a2.cin == 32 && a2.k_pad == 288 && a2.cout == 64 && a2.cout_pad >= 64 &&
                    a2.up2 == 0 && a2.chain_w == nullptr && a2.split == 0 && a2.res == nullptr && a2.act == YMI_ACT_SILU,
                "%s: the second descriptor must be Conv(32, 64, k=3, s=2, p=1) with SiLU (no residual / split / chain)", who);
    YMI_REQUIRE(a2.n == a1.n && a2.h == a1.ho && a2.w_in == a1.wo && a2.ho == (a2.h - 1) / 2 + 1 && a2.wo == (a2.w_in - 1) / 2 + 1,
                "%s: the second convolution must read the first one's output (%dx%d), got %dx%d", who, a1.ho, a1.wo, a2.h, a2.w_in);
    YMI_REQUIRE(((int64_t)a2.M + 1) * a2.y_cs < ((int64_t)1 << 31), "%s: output tensor too large for 32-bit offsets", who);
    return YMI_OK;
}

// a1: the stem in its super-pixel form (as for conv_stem_planar_launch; y is not written); a2: Conv(32, 64, 3, 2, 1) over the stem's output (x is not read)
int stem_body1_planar_launch(const ConvArgs& a1, const ConvArgs& a2, const void* const* imgs, int dtype, hipStream_t s) {
    { const int rc = stem_body1_check(a1, a2, "ymi_stem_body1_planar"); if (rc != YMI_OK) return rc; }
    YMI_REQUIRE((2 * a1.w_in) % 8 == 0, "ymi_stem_body1_planar: the image width must be a multiple of 8");
    return dtype == YMI_F16 ? stem_body1_launch_t<YMI_F16, true>(a1, a2, imgs, s) : stem_body1_launch_t<YMI_BF16, true>(a1, a2, imgs, s);
}

// The same pair fed from the NHWC4 canvas a1.x (n, H, W / 2, 8) the letterbox wrote (dynamic-shape streams)
int stem_body1_launch(const ConvArgs& a1, const ConvArgs& a2, int dtype, hipStream_t s) {
    { const int rc = stem_body1_check(a1, a2, "ymi_stem_body1"); if (rc != YMI_OK) return rc; }
    YMI_REQUIRE(a1.x != nullptr && a1.x_cs == 8, "ymi_stem_body1: the stem must read the NHWC4 canvas (8 halves per super-pixel)");
    YMI_REQUIRE((int64_t)a1.n * a1.h * a1.w_in * 8 < ((int64_t)1 << 31), "ymi_stem_body1: input canvas too large for 32-bit offsets");
    return dtype == YMI_F16 ? stem_body1_launch_t<YMI_F16, false>(a1, a2, nullptr, s) : stem_body1_launch_t<YMI_BF16, false>(a1, a2, nullptr, s);
}

}  // namespace ymi
