// YOLOv5 r6.0 stem: Conv(3, c, k=6, s=2, p=2) + BN + SiLU (reference yolort/models/darknetv6.py:81,
// yolort/v5/models/common.py:69-70) over the NHWC4 (RGB0) letterboxed batch.
//
// The plan presents the stem as a 6x3, stride (2,1), pad (2,1) convolution over width-halved
// "super-pixels" of 8 channels (2 pixels x RGB0, 16 bytes).  The implicit-GEMM kernel runs it at the
// LDS-fill limit (every super-pixel is fetched 9 times).  This kernel instead gives each block an
// 8 x 32 output tile of one image: its (2*8+4) x (32+2) super-pixel input patch (10.9 KiB) is DMA'd into
// LDS once, the whole folded weight matrix (<= 64 x 144) sits in registers, and the 18 taps read their
// MFMA fragments from the patch at constant offsets (16-byte lane stride: conflict free).
// K = 18 taps x 8 = 144 = 9 MFMA k16 steps (lanes 0-31 take tap 2s, lanes 32-63 tap 2s+1).
#include "conv_common.hpp"

namespace ymi {

constexpr int STH = 8, STW = 32;                 // output tile
constexpr int SPH = 2 * STH + 4, SPW = STW + 2;  // input patch: 20 rows x 34 super-pixels
constexpr int SPIECES = 11;                      // ceil(20*34 / 64) DMA pieces of 64 super-pixels (1 KiB)

template <int DT, int ODT, int TN>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    typedef typename Mfma<DT>::frag frag;
    __shared__ __attribute__((aligned(16))) uint16_t patch[12 * 512];   // 12 KiB (680 super-pixels used)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = a.nblk_m;
    int t = xcd_remap(blockIdx.x, nblk);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int img = t / tiles_y;
    const int oy0 = ty * STH, ox0 = tx * STW;

    f32x4 bias_regs[TN][4];
    load_bias<TN>(a, 0, lane >> 5, bias_regs);
    // ---- patch: super-pixel q -> (row q/34, col q%34) -> input (2*oy0 - 2 + row, ox0 - 1 + col) ----
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int pi = wave * 3 + j;
        pi = pi < SPIECES ? pi : SPIECES - 1;
        const int q = pi * 64 + lane;
        const int qc = q < SPH * SPW ? q : SPH * SPW - 1;
        const int pr = qc / SPW, pc = qc - pr * SPW;
        const int iy = 2 * oy0 - 2 + pr, ix = ox0 - 1 + pc;
        const bool ok = (q < SPH * SPW) && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
        const int off = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs : a.x_zero_off;
        glds16(a.x + off, patch + pi * 512);
    }
    // ---- weights: all 9 k16-steps of this lane's cout row(s) into registers (L2-resident, 16 B loads) ----
    frag wf[TN][9];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const uint16_t* wr = a.w + (int64_t)(i * 32 + (lane & 31)) * a.k_pad + 8 * (lane >> 5);
#pragma unroll
        for (int s = 0; s < 9; ++s) wf[i][s] = *reinterpret_cast<const frag*>(wr + 16 * s);
    }
    f32x16 acc[TN][2];
    init_acc<TN, 2>(acc, bias_regs);   // accumulate on top of the bias

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // wave w owns output rows 2w, 2w+1 of the tile (two groups of 32 pixels)
    const int hi = lane >> 5, px = lane & 31;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        // tap = 2s + hi -> (ky, kx') = (tap / 3, tap % 3): compile-time per half
        const int tap0 = 2 * s, tap1 = 2 * s + 1;
        const int o0 = (tap0 / 3) * SPW + (tap0 % 3), o1 = (tap1 / 3) * SPW + (tap1 % 3);
        const int toff = hi ? o1 : o0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 2 * (2 * wave + j);   // patch row of output row (2w + j) at ky = 0
            const frag af = *reinterpret_cast<const frag*>(patch + (row * SPW + px + toff) * 8);
#pragma unroll
            for (int i = 0; i < TN; ++i) acc[i][j] = Mfma<DT>::run(wf[i][s], af, acc[i][j]);
        }
    }

    // ---- epilogue: SiLU, 16-byte stores straight from the MFMA layout (conv_common.hpp) ----
    finish_wave_tile<DT, ODT, TN, 2>(a, acc, 0, hi, [&](int j, int64_t& m, bool& ok) {
        const int oy = oy0 + 2 * wave + j, ox = ox0 + px;
        ok = oy < a.ho && ox < a.wo;
        m = ((int64_t)img * a.ho + oy) * a.wo + ox;
    });
}

// -------------------------------------------------------------------------------------------------------------
// Stem fed straight from the PLANAR input images (identity-size batches: every image already is the (3, H, W) canvas,
// so the reference's resize is the identity and batch_images pads nothing, transform.py:53-97,297-330).  The NHWC4
// round trip of the letterbox (8 B written + 8 B read per pixel) disappears: the block DMAs its 20-row x 80-pixel
// patch of each of the three planes into LDS (rows start at a 16-byte aligned pixel: 2*ox0 - 8) and gathers the
// super-pixel fragments from there -- pixels (2c, 2c+1) of a plane are one aligned 4-byte LDS read, three reads and
// four byte-permutes rebuild the [R G B 0 | R G B 0] fragment the MFMA expects.  Same weights, same K order, same
// epilogue: results are bit-identical to letterbox + conv_stem_kernel.
// -------------------------------------------------------------------------------------------------------------
constexpr int PL_MAX = 32;                       // images per launch (pointers travel as kernel arguments)
constexpr int PPW = 80;                          // patch pixels per row (10 lanes x 16 B)
constexpr int PL_LANES = 3 * SPH * (PPW / 8);    // 600 16-byte pieces-of-row = 10 DMA instructions

struct PlanarArgs {
    const uint16_t* img[PL_MAX];
};

// Persistent form (round 2): the first version gave every 8 x 32 tile its own block, and each block paid the patch DMA
// latency, a private copy of the weights from L2 (36 KiB per block against a 9.6 KiB patch) and the store acknowledgements
// of its epilogue in sequence -- 116 us per 32-image batch at 640 x 640 against a 46 us HBM bound.  Now a block keeps the
// weights in registers and walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the patch of tile i+1 is DMA'd into the
// other LDS buffer while tile i is computed and stored.
template <int DT, int ODT, int TN>
__global__ __launch_bounds__(256, 2) void conv_stem_planar_kernel(const ConvArgs a, const PlanarArgs pl, int tiles_x, int tiles_y) {
    typedef typename Mfma<DT>::frag frag;
    __shared__ __attribute__((aligned(16))) uint16_t patch2[2][12 * 512];   // 2 x [3 planes][20 rows][80 px] (9600 B used of each)
    __shared__ __attribute__((aligned(16))) uint16_t wl[TN * 9 * 64 * 8];   // weights in fragment order: [(i, s)][lane] x 16 B (registers would hold them across the epilogue: 2 waves / SIMD)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = a.nblk_m;
    const int H = a.h, W = 2 * a.w_in;           // image size in pixels (the conv geometry counts super-pixels)

    // ---- patch DMA: entry e = (plane, row, seg) -> 8 pixels starting at (2*oy0 - 2 + row, 2*ox0 - 8 + 8*seg); the per-lane
    //      (plane, row, seg) never changes, only the tile origin does ----
    int e_off[3];        // element offset of the lane's entry relative to the tile origin, or -1 (no entry: zero page)
    int e_row[3], e_col[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pi = wave * 3 + j;
        const int e = pi * 64 + lane;
        const int ec = e < PL_LANES ? e : PL_LANES - 1;
        const int plane = ec / (SPH * (PPW / 8));
        const int rem = ec - plane * (SPH * (PPW / 8));
        const int pr = rem / (PPW / 8), seg = rem - pr * (PPW / 8);
        e_row[j] = pr - 2;
        e_col[j] = 8 * seg - 8;
        e_off[j] = e < PL_LANES ? plane : -1;
    }
    auto tile_origin = [&](int idx, int& img, int& oy0, int& ox0) {
        int t = xcd_remap(idx, nblk);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        img = t / tiles_y;
        oy0 = ty * STH;
        ox0 = tx * STW;
    };
    auto issue_patch = [&](int idx, uint16_t* dst) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        const uint16_t* base = pl.img[img];      // wave-uniform index into the kernel arguments: one scalar load
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int pi = wave * 3 + j;
            if (pi * 64 < PL_LANES) {            // wave-uniform
                const int iy = 2 * oy0 + e_row[j], ix = 2 * ox0 + e_col[j];
                const bool ok = (e_off[j] >= 0) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);   // W % 8 == 0: a segment is in or out as a whole
                const uint16_t* src = ok ? base + ((int64_t)e_off[j] * H + iy) * W + ix : a.zeros;
                glds16(src, dst + pi * 512);
            }
        }
    };
    int idx = blockIdx.x;
    if (idx < nblk) issue_patch(idx, patch2[0]);
    // ---- weights: all 9 k16-steps of every cout row into LDS, ONCE per block (wave w copies the steps s = w, w+4, ...) ----
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const uint16_t* wr = a.w + (int64_t)(i * 32 + (lane & 31)) * a.k_pad + 8 * (lane >> 5);
        for (int s = wave; s < 9; s += 4)
            *reinterpret_cast<frag*>(wl + ((i * 9 + s) * 64 + lane) * 8) = *reinterpret_cast<const frag*>(wr + 16 * s);
    }
    // wave w owns output rows 2w, 2w+1 of the tile (two groups of 32 pixels)
    const int hi = lane >> 5, px = lane & 31;
    constexpr int PLANE_HALFS = SPH * PPW;
    int buf = 0;
    for (; idx < nblk; idx += gridDim.x) {
        // this tile's patch (issued one tile ago) has landed, and so have the stores of the previous tile; after the barrier
        // nobody reads the other buffer any more: it takes the next tile's patch while this one is computed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (idx + (int)gridDim.x < nblk) issue_patch(idx + gridDim.x, patch2[buf ^ 1]);
        const uint16_t* patch = patch2[buf];
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);

        f32x4 bias_regs[TN][4];
        load_bias<TN>(a, 0, lane >> 5, bias_regs);
        f32x16 acc[TN][2];
        init_acc<TN, 2>(acc, bias_regs);   // accumulate on top of the bias
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            frag wf[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const frag*>(wl + ((i * 9 + s) * 64 + lane) * 8);
            // tap = 2s + hi -> (ky, kx') = (tap / 3, tap % 3): compile-time per half; super-pixel column c = px + kx' sits at
            // pixels 2*(ox0 - 1 + c) .. +1 = patch pixels 2c + 6, 2c + 7
            const int tap0 = 2 * s, tap1 = 2 * s + 1;
            const int o0 = (tap0 / 3) * PPW + 2 * (tap0 % 3), o1 = (tap1 / 3) * PPW + 2 * (tap1 % 3);
            const int toff = hi ? o1 : o0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 2 * (2 * wave + j);   // patch row of output row (2w + j) at ky = 0
                const uint16_t* p0 = patch + row * PPW + 2 * px + 6 + toff;
                const uint32_t r = *reinterpret_cast<const uint32_t*>(p0);
                const uint32_t g = *reinterpret_cast<const uint32_t*>(p0 + PLANE_HALFS);
                const uint32_t b = *reinterpret_cast<const uint32_t*>(p0 + 2 * PLANE_HALFS);
                u32x4 q;
                q[0] = (r & 0xffffu) | (g << 16);        // R0 G0
                q[1] = b & 0xffffu;                      // B0 0
                q[2] = (r >> 16) | (g & 0xffff0000u);    // R1 G1
                q[3] = b >> 16;                          // B1 0
                frag af;
                __builtin_memcpy(&af, &q, 16);
#pragma unroll
                for (int i = 0; i < TN; ++i) acc[i][j] = Mfma<DT>::run(wf[i], af, acc[i][j]);
            }
        }

        finish_wave_tile<DT, ODT, TN, 2>(a, acc, 0, hi, [&](int j, int64_t& m, bool& ok) {
            const int oy = oy0 + 2 * wave + j, ox = ox0 + px;
            ok = oy < a.ho && ox < a.wo;
            m = ((int64_t)img * a.ho + oy) * a.wo + ox;
        });
        buf ^= 1;
    }
}

template <int DT, int ODT>
static int stem_planar_launch_t(const ConvArgs& a0, const void* const* imgs, hipStream_t s) {
    const int tiles_x = cdiv(a0.wo, STW), tiles_y = cdiv(a0.ho, STH);
    for (int base = 0; base < a0.n; base += PL_MAX) {
        ConvArgs a = a0;
        a.n = a0.n - base < PL_MAX ? a0.n - base : PL_MAX;
        PlanarArgs pl;
        for (int i = 0; i < PL_MAX; ++i) pl.img[i] = (const uint16_t*)imgs[base + (i < a.n ? i : 0)];
        // outputs (and the residual-free epilogue) of this group start at image `base`
        const int64_t ystep = (int64_t)base * a0.ho * a0.wo * a0.y_cs;
        a.y = ODT == YMI_F32 ? (void*)((float*)a0.y + ystep) : (void*)((uint16_t*)a0.y + ystep);
        a.nblk_m = a.n * tiles_x * tiles_y;
        a.nblk_n = 1;
        a.M = a.n * a.ho * a.wo;
        // persistent blocks: as many as are resident at once (registers: 3 per CU at cout <= 32, 2 above); 256 CUs, a multiple
        // of 8 so that a block stays on one XCD's tile range
        auto launch = [&](auto kfn) {
            static int per_cu = 0;
            if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, 0) != hipSuccess || per_cu < 1)) per_cu = 2;
            const int resident = per_cu * 256;
            dim3 grid(a.nblk_m < resident ? a.nblk_m : resident), block(256);
            hipLaunchKernelGGL(kfn, grid, block, 0, s, a, pl, tiles_x, tiles_y);
        };
        if (a.cout_pad <= 32) launch(conv_stem_planar_kernel<DT, ODT, 1>);
        else launch(conv_stem_planar_kernel<DT, ODT, 2>);
    }
    return check_launch("conv_stem_planar_kernel");
}

int conv_stem_planar_launch(const ConvArgs& a, const void* const* imgs, int dtype, int out_dtype, hipStream_t s) {
    YMI_REQUIRE(a.cin == 8 && a.kh == 6 && a.kw == 3 && a.sh == 2 && a.sw == 1 && a.ph == 2 && a.pw == 1 && a.k_pad >= 144 && a.cout_pad <= 64 &&
                    a.zeros != nullptr && a.split == 0 && a.res == nullptr,
                "ymi_conv_stem_planar: the stem must be in its 6x3 s(2,1) p(2,1) super-pixel form with cout <= 64 (and needs desc.zeros)");
    YMI_REQUIRE((2 * a.w_in) % 8 == 0, "ymi_conv_stem_planar: the image width must be a multiple of 8");
    if (dtype == YMI_F16) return out_dtype == YMI_F32 ? stem_planar_launch_t<YMI_F16, YMI_F32>(a, imgs, s) : stem_planar_launch_t<YMI_F16, YMI_F16>(a, imgs, s);
    return out_dtype == YMI_F32 ? stem_planar_launch_t<YMI_BF16, YMI_F32>(a, imgs, s) : stem_planar_launch_t<YMI_BF16, YMI_BF16>(a, imgs, s);
}

template <int DT, int ODT>
static int stem_launch_t(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int tiles_x = cdiv(a.wo, STW), tiles_y = cdiv(a.ho, STH);
    a.nblk_m = a.n * tiles_x * tiles_y;
    a.nblk_n = 1;
    dim3 grid(a.nblk_m), block(256);
    if (a.cout_pad <= 32) hipLaunchKernelGGL((conv_stem_kernel<DT, ODT, 1>), grid, block, 0, s, a, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_stem_kernel<DT, ODT, 2>), grid, block, 0, s, a, tiles_x, tiles_y);
    return check_launch("conv_stem_kernel");
}

int conv_stem_launch(const ConvArgs& a, int dtype, int out_dtype, hipStream_t s) {
    YMI_REQUIRE(a.cin == 8 && a.kh == 6 && a.kw == 3 && a.sh == 2 && a.sw == 1 && a.ph == 2 && a.pw == 1 && a.x_cs == 8 && a.k_pad >= 144 &&
                    a.cout_pad <= 64 && a.zeros != nullptr && a.split == 0 && a.res == nullptr,
                "ymi_conv2d: the stem kernel handles the 6x3 s(2,1) p(2,1) super-pixel form with cout <= 64 only");
    if (dtype == YMI_F16) return out_dtype == YMI_F32 ? stem_launch_t<YMI_F16, YMI_F32>(a, s) : stem_launch_t<YMI_F16, YMI_F16>(a, s);
    return out_dtype == YMI_F32 ? stem_launch_t<YMI_BF16, YMI_F32>(a, s) : stem_launch_t<YMI_BF16, YMI_BF16>(a, s);
}

}  // namespace ymi
