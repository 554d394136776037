// 3x3 STRIDE-2 convolution of a 64-channel input into 128 channels (gfx950), weights in REGISTERS: tile 134 (round 4).
//
// The down-sampling convolutions of the backbone / PAN (darknetv6.py:85-96, path_aggregation_network.py:140-156) ran on the generic implicit GEMM at
// 0.16-0.32 of their bounds (profiles/r03z_layer_table_c2_pmc.csv: body.3 of yolov5s, 64 -> 128 at 160^2 -> 80^2, 62 us against an HBM bound of 19.7 us).
// This is the design of conv3x3_rw.hip (tile 133) at stride 2:
//   * a block is 4 waves on ONE 8 x 8 output tile; wave w owns the 32 couts [32 w, 32 w + 32) of all 64 pixels and keeps the 36 weight fragments of its
//     cout group in registers for the whole kernel (144 VGPRs), so the LDS holds nothing but two 17 x 17-pixel input patches per block (2 x 39 KiB) and TWO
//     blocks are resident per CU: one block's barrier, DMA issue and SiLU epilogue run under the other's MFMAs;
//   * the patch keeps a row's columns split by parity ([even columns | odd columns], like conv3x3_c32.hip at stride 2): the 32 lanes of a fragment read then
//     touch consecutive 128-byte slots.  Chunk swizzle v = ((row >> 1) & 3) << 1 | ((column index >> 1) & 1) on top of the slot parity: each of
//     ds_read_b128's four 16-lane groups ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md, LDS) covers 4 tile rows x 4 consecutive columns, i.e. every
//     (slot parity, v) pair once: conflict-free for all nine taps;
//   * persistent blocks, the patch of tile i+1 DMA'd while tile i computes, one barrier per tile.
// K order (tap-major, channel-minor), accumulator layout and lean epilogue of the other conv kernels: bit-identical to the implicit GEMM (tiles 111 / 112).
// Replaces yolort/v5/models/common.py:69-70 for Conv(64, 128, 3, 2).
#include "conv_common.hpp"
#include <cstdlib>

namespace ymi {

constexpr int R2_T = 8;                                    // output tile: 8 x 8 pixels, two 32-pixel groups (tile rows 0-3, 4-7)
constexpr int R2_PH = 2 * R2_T + 1;                        // 17 patch rows
constexpr int R2_PITCH = 18, R2_HO = 10;                   // row pitch in slots: even input columns (index 0..8) at 0..8, odd ones (0..7) at 10..17, slot 9 unused
constexpr int R2_SLOTS = R2_PH * R2_PITCH;                 // 306 pixel slots of 128 B
constexpr int R2_PIECES = (R2_SLOTS * 8 + 63) / 64;        // 39 DMA pieces of 1 KiB
constexpr int R2_PPW = (R2_PIECES + 3) / 4;                // 10 pieces per wave
constexpr int R2_PATCH_BYTES = R2_PIECES * 1024;
constexpr int R2_J1 = 8 * R2_PITCH * 128;                  // byte distance of a wave's second pixel group: four tile rows = eight patch rows down (the swizzle repeats)
constexpr int R2_BIAS_BYTES = 512;                         // bias [4 cout groups][4 octets][2 halves] x 16 B

__device__ __forceinline__ int r2_swz(int pr, int ci) { return (((pr >> 1) & 3) << 1) | ((ci >> 1) & 1); }

// (Round 4 also built this kernel with a dedicated DMA wave and three patch buffers -- tile 136, compute waves that never wait for memory: 57 us against 50 on
// yolov5s' body.3, profiles/r04e_tile136_first.txt; removed in round 5.)
template <int DT>
__global__ __launch_bounds__(256, 2) void conv3x3_rw2_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles) {
    typedef typename Mfma<DT>::frag frag;
    constexpr int CIN = 64, KC = CIN / 16, NU = 9 * KC;    // (tap, k16) units: one weight fragment, two activation fragments, two MFMAs each
    extern __shared__ __attribute__((aligned(16))) unsigned char r2_sm[];
    f32x4* bl = reinterpret_cast<f32x4*>(r2_sm);
    unsigned char* patch0 = r2_sm + R2_BIAS_BYTES;          // two patch buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = cout group
    const int hi = lane >> 5, frow = lane & 31;

    auto tile_origin = [&](int idx, int& img, int& oy0, int& ox0) {
        int t = xcd_remap(idx, ntiles);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        img = t / tiles_y;
        oy0 = ty * R2_T;
        ox0 = tx * R2_T;
    };

    // ---- this wave's weights: fragment (tap, kc) = rows wave*32 + frow, k = tap*64 + kc*16 + hi*8 .. +7 ----
    frag wf[NU];
    {
        const uint16_t* wr = a.w + (int64_t)(wave * 32 + frow) * a.k_pad + hi * 8;
#pragma unroll
        for (int u = 0; u < NU; ++u) wf[u] = *reinterpret_cast<const frag*>(wr + u * 16);
    }
    if (tid < 32) {   // bias quad of (group t, octet g, half h): couts t*32 + g*8 + h*4 ..
        const int t = tid >> 3, g = (tid >> 1) & 3, h = tid & 1;
        bl[tid] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + g * 8 + h * 4);
    }

    // ---- fragment geometry (fixed per lane): pixel group 0 of the tile: p = frow -> (r, c) = (frow >> 3, frow & 7); group 1 is four tile rows below ----
    const int pr_o = frow >> 3, pc_o = frow & 7;
    int ea[9];            // byte offset of chunk `hi` (k16 step 0) of the tap's pixel; step kc: ^ (kc << 5)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        const int pr = 2 * pr_o + dy, col = 2 * pc_o + dx;
        const int ci = col >> 1;
        ea[t] = (pr * R2_PITCH + ((col & 1) ? R2_HO : 0) + ci) * 128 + ((hi ^ r2_swz(pr, ci)) * 16);
    }

    // the patch pieces of this wave (tile 134): entry e = piece*64 + lane -> slot e >> 3 = (pr, sc), position e & 7 holds chunk pos ^ v(pr, ci); fixed per lane
    int p_rc[R2_PPW];       // pr << 16 | chunk << 8 | column (relative to the patch origin), or -1: nothing to fetch; the source offset is rebuilt from it per tile
                          // (four vector instructions per piece: the explicitly scheduled fragment loop needs the ten registers a second array would take)
#pragma unroll
    for (int j = 0; j < R2_PPW; ++j) {
        int pi = wave * R2_PPW + j;
        pi = pi < R2_PIECES ? pi : R2_PIECES - 1;          // surplus slots re-send the last piece (identical bytes)
        const int e = pi * 64 + lane;
        const int q = e >> 3;
        const int qc = q < R2_SLOTS ? q : R2_SLOTS - 1;
        const int pr = qc / R2_PITCH, sc = qc - pr * R2_PITCH;
        const int odd = sc >= R2_HO ? 1 : 0;
        const int ci = odd ? sc - R2_HO : sc;
        const int col = 2 * ci + odd;                      // 0 .. 16 (slot 9: ci = 9 -> col 18, past the patch)
        const int chunk = (e & 7) ^ r2_swz(pr, ci);
        p_rc[j] = (q < R2_SLOTS && col <= 2 * R2_T) ? ((pr << 16) | (chunk << 8) | col) : -1;
    }
    auto issue_patch = [&](int idx, unsigned char* dst) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
        const int base = ((img * a.h + iy0) * a.w_in + ix0) * a.x_cs;
#pragma unroll
        for (int j = 0; j < R2_PPW; ++j) {
            int pi = wave * R2_PPW + j;
            pi = pi < R2_PIECES ? pi : R2_PIECES - 1;
            const int pr = p_rc[j] >> 16, col = p_rc[j] & 0xff;
            const bool ok = p_rc[j] >= 0 && ((unsigned)(iy0 + pr) < (unsigned)a.h) && ((unsigned)(ix0 + col) < (unsigned)a.w_in);
            const int off = ok ? base + (pr * a.w_in + col) * a.x_cs + ((p_rc[j] >> 8) & 0xff) * 8 : a.x_zero_off;
            glds16(a.x + off, reinterpret_cast<uint16_t*>(dst + pi * 1024));
        }
    };

    int idx = blockIdx.x;
    int buf = 0;
    if (idx < ntiles) issue_patch(idx, patch0);
    for (; idx < ntiles; idx += gridDim.x) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // patch i has landed; everyone is done reading patch i-1 (first pass: the bias is written)
        const unsigned char* pb = patch0 + buf * R2_PATCH_BYTES;
        if (idx + (int)gridDim.x < ntiles && !(a.debug & 2)) issue_patch(idx + gridDim.x, patch0 + (buf ^ 1) * R2_PATCH_BYTES);   // (debug & 2, tile id + 0x200: ablation without the patch loads)
        buf ^= 1;

        if (a.debug & 16) continue;   // ablation (tile id + 0x1000): the patch traffic alone -- no fragment reads, no MFMAs, no stores
        f32x16 acc[1][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = bl[(wave * 4 + g) * 2 + hi];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][0][g * 4 + e] = acc[0][1][g * 4 + e] = b[e];
        }
        // unit = (tap, k16): its two activation fragments are fetched under the previous unit's MFMAs; the tap's base address is laundered inside the loop (left alone,
        // the compiler hoists all 9 * KC addresses out of the tile loop: 36 registers this kernel does not have)
        // (Explicitly scheduled fragment reads -- inline-assembly ds_read_b128 two to four units ahead of their MFMAs with counted lgkmcnt -- were measured in round 4:
        // neither the compute-only time nor the kernel moved, two waves per SIMD already cover each other's LDS round trips; profiles/r04g_*, r04i_*.  Removed in round 5.)
        frag fa[2][2];
        {
            auto read_plain = [&](auto ut, auto bt) {
                constexpr int u = decltype(ut)::value, b = decltype(bt)::value;
                constexpr int t = u / KC, kc = u % KC;
                int eb = ea[t];
                asm volatile("" : "+v"(eb));
                const unsigned char* p0 = pb + (eb ^ (kc << 5));
                fa[b][0] = *reinterpret_cast<const frag*>(p0);
                fa[b][1] = *reinterpret_cast<const frag*>(p0 + R2_J1);
            };
            read_plain(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            static_for<0, NU>([&](auto ut) {
                constexpr int u = decltype(ut)::value;
                if constexpr (u + 1 < NU) read_plain(std::integral_constant<int, u + 1>{}, std::integral_constant<int, (u + 1) & 1>{});
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][j] = Mfma<DT>::run(wf[u], fa[u & 1][j], acc[0][j]);
            });
        }
        // the lean epilogue only (the launcher admits nothing else: SiLU, cout = 128, tensors below 2^31 elements)
        auto pix = [&](int j, int64_t& m, bool& ok) {
            const int oy = oy0 + pr_o + 4 * j, ox = ox0 + pc_o;
            ok = oy < a.ho && ox < a.wo;
            m = ((int64_t)img * a.ho + oy) * a.wo + ox;
        };
        if (a.debug & 8) {   // ablation (tile id + 0x800): no epilogue, no stores -- results are garbage
            asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[0][1][0]));
            continue;
        }
        finish_wave_tile_lean<DT, 1, 2, false>(a, acc, wave * 32, hi, pix);
    }
}

template <int DT>
static int launch_rw2(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int tiles_x = cdiv(a.wo, R2_T), tiles_y = cdiv(a.ho, R2_T);
    const int ntiles = a.n * tiles_x * tiles_y;
    const size_t lds = R2_BIAS_BYTES + (size_t)2 * R2_PATCH_BYTES;
    auto kfn = conv3x3_rw2_kernel<DT>;
    if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    int resident = 512;   // two 4-wave blocks per CU
    if (const char* e = getenv("YOLORT_AMD_RES3X3_BLOCKS")) {   // test aid: few blocks walk many tiles (the persistent loop on small inputs)
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) resident = v;
    }
    a.nblk_m = ntiles;
    a.nblk_n = 1;
    hipLaunchKernelGGL(kfn, dim3(ntiles < resident ? ntiles : resident), dim3(256), lds, s, a, tiles_x, tiles_y, ntiles);
    return check_launch("conv3x3_rw2_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Tile 135: the same design for cin = 128 (K = 1152), cout = 128 / 256 -- pan.layer_blocks.1 and backbone.body.5 of yolov5s (36 / 50 us against 8 / 12 us).
// A wave cannot hold 72 weight fragments, so K is SPLIT over two waves: a block is 8 waves = 4 cout groups x 2 channel halves on one 8 x 8 output tile (cout
// group g = couts [32 g, 32 g + 32) of the block's 128; half kh = input channels [64 kh, 64 kh + 64) of every tap: 36 fragments = 144 VGPRs per wave).  The
// kh = 1 waves hand their partial sums to the kh = 0 waves through the (consumed) patch buffer; those add them, apply SiLU and store.  cout = 256: two blocks
// per tile (blockIdx.y), each fetching the patch (the second one from L2).  One 17 x 17 patch of 256-byte slots is 74 KiB: double-buffered, ONE block per CU
// (8 waves = 2 per SIMD).  Chunk swizzle over all 16 positions of a slot: v = ((row >> 1) & 3) << 2 | (column index & 3) -- a 256-byte slot is a whole
// bank row, so the 16 lanes of a ds_read_b128 group (4 tile rows x 4 consecutive columns) must take 16 different positions.
// The split sum rounds differently from the implicit GEMM's single K loop in the last bit: NOT bit-identical to tiles 111 / 115 (within the fp32
// summation-order noise every tile choice has; the per-launch parity test bounds it against the fp32 oracle).
constexpr int R5_PITCH = 18, R5_HO = 10;
constexpr int R5_SLOTS = R2_PH * R5_PITCH;                 // 306 pixel slots of 256 B
constexpr int R5_PIECES = (R5_SLOTS * 16 + 63) / 64;       // 77 DMA pieces of 1 KiB
constexpr int R5_PPW = (R5_PIECES + 7) / 8;                // 10 pieces per wave
constexpr int R5_PATCH_BYTES = R5_PIECES * 1024;           // 78848
constexpr int R5_J1 = 8 * R5_PITCH * 256;

__device__ __forceinline__ int r5_swz(int pr, int ci) { return (((pr >> 1) & 3) << 2) | (ci & 3); }

template <int DT>
__global__ __launch_bounds__(512, 1) void conv3x3_rw3_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles) {
    typedef typename Mfma<DT>::frag frag;
    constexpr int KC = 4, NU = 9 * KC;                      // per wave: (tap, k16 step of its 64-channel half)
    extern __shared__ __attribute__((aligned(16))) unsigned char r5_sm[];
    f32x4* bl = reinterpret_cast<f32x4*>(r5_sm);
    unsigned char* patch0 = r5_sm + R2_BIAS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;
    const int ct = wave & 3, kh = wave >> 2;               // cout group of the block's 128, channel half
    const int cbase = blockIdx.y * 128;                     // this block's couts

    frag wf[NU];
    {
        const uint16_t* wr = a.w + (int64_t)(cbase + ct * 32 + frow) * a.k_pad + kh * 64 + hi * 8;
#pragma unroll
        for (int u = 0; u < NU; ++u) wf[u] = *reinterpret_cast<const frag*>(wr + (u / KC) * 128 + (u % KC) * 16);
    }
    if (tid < 32) {
        const int t = tid >> 3, g = (tid >> 1) & 3, h = tid & 1;
        bl[tid] = *reinterpret_cast<const f32x4*>(a.bias + cbase + t * 32 + g * 8 + h * 4);
    }

    int p_rc[R5_PPW];     // pr << 16 | chunk << 8 | column, or -1 (the source offset is rebuilt from it per tile: ten registers this kernel does not have)
#pragma unroll
    for (int j = 0; j < R5_PPW; ++j) {
        int pi = wave * R5_PPW + j;
        pi = pi < R5_PIECES ? pi : R5_PIECES - 1;
        const int e = pi * 64 + lane;
        const int q = e >> 4;
        const int qc = q < R5_SLOTS ? q : R5_SLOTS - 1;
        const int pr = qc / R5_PITCH, sc = qc - pr * R5_PITCH;
        const int odd = sc >= R5_HO ? 1 : 0;
        const int ci = odd ? sc - R5_HO : sc;
        const int col = 2 * ci + odd;
        const int chunk = (e & 15) ^ r5_swz(pr, ci);
        p_rc[j] = (q < R5_SLOTS && col <= 2 * R2_T) ? ((pr << 16) | (chunk << 8) | col) : -1;
    }
    const int pr_o = frow >> 3, pc_o = frow & 7;
    int ea[9];            // byte offset of chunk (kh*8 + hi) (k16 step 0 of this wave's half) of the tap's pixel; step kc: ^ (kc << 5)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        const int pr = 2 * pr_o + dy, col = 2 * pc_o + dx;
        const int ci = col >> 1;
        ea[t] = (pr * R5_PITCH + ((col & 1) ? R5_HO : 0) + ci) * 256 + (((kh * 8 + hi) ^ r5_swz(pr, ci)) * 16);
    }

    auto tile_origin = [&](int idx, int& img, int& oy0, int& ox0) {
        int t = xcd_remap(idx, ntiles);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        img = t / tiles_y;
        oy0 = ty * R2_T;
        ox0 = tx * R2_T;
    };
    auto issue_patch = [&](int idx, unsigned char* dst) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
#pragma unroll
        for (int j = 0; j < R5_PPW; ++j) {
            int pi = wave * R5_PPW + j;
            pi = pi < R5_PIECES ? pi : R5_PIECES - 1;
            const int iy = iy0 + (p_rc[j] >> 16), ix = ix0 + (p_rc[j] & 0xff);
            const bool ok = p_rc[j] >= 0 && ((unsigned)iy < (unsigned)a.h) && ((unsigned)ix < (unsigned)a.w_in);
            const int off = ok ? ((img * a.h + iy) * a.w_in + ix) * a.x_cs + ((p_rc[j] >> 8) & 0xff) * 8 : a.x_zero_off;
            glds16(a.x + off, reinterpret_cast<uint16_t*>(dst + pi * 1024));
        }
    };

    int idx = blockIdx.x;
    int buf = 0;
    if (idx < ntiles) issue_patch(idx, patch0);
    for (; idx < ntiles; idx += gridDim.x) {
        int img, oy0, ox0;
        tile_origin(idx, img, oy0, ox0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // patch i has landed; the kh = 0 waves are done reading the partial sums out of patch i-1's buffer
        unsigned char* pb = patch0 + buf * R5_PATCH_BYTES;
        if (idx + (int)gridDim.x < ntiles) issue_patch(idx + gridDim.x, patch0 + (buf ^ 1) * R5_PATCH_BYTES);
        buf ^= 1;

        f32x16 acc[1][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 b = bl[(ct * 4 + g) * 2 + hi];
            if (kh) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; b = z; }   // the bias starts the kh = 0 sum only
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][0][g * 4 + e] = acc[0][1][g * 4 + e] = b[e];
        }
        frag fa[2][2];
        auto read_unit = [&](auto ut, auto bt) {
            constexpr int u = decltype(ut)::value, b = decltype(bt)::value;
            constexpr int t = u / KC, kc = u % KC;
            int eb = ea[t];
            asm volatile("" : "+v"(eb));
            const unsigned char* p0 = pb + (eb ^ (kc << 5));
            fa[b][0] = *reinterpret_cast<const frag*>(p0);
            fa[b][1] = *reinterpret_cast<const frag*>(p0 + R5_J1);
        };
        read_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, NU>([&](auto ut) {
            constexpr int u = decltype(ut)::value;
            if constexpr (u + 1 < NU) read_unit(std::integral_constant<int, u + 1>{}, std::integral_constant<int, (u + 1) & 1>{});
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[0][j] = Mfma<DT>::run(wf[u], fa[u & 1][j], acc[0][j]);
        });
        __syncthreads();   // every wave is done reading patch i: its buffer carries the kh = 1 partial sums to the kh = 0 waves
        f32x4* red = reinterpret_cast<f32x4*>(pb) + (ct * 8) * 64 + lane;   // [cout group][8 quads of (j, g)][64 lanes] x 16 B: conflict-free
        if (kh) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[0][j][g * 4], acc[0][j][g * 4 + 1], acc[0][j][g * 4 + 2], acc[0][j][g * 4 + 3]};
                    red[(j * 4 + g) * 64] = v;
                }
        }
        __syncthreads();
        if (!kh) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = red[(j * 4 + g) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0][j][g * 4 + e] += v[e];
                }
            auto pix = [&](int j, int64_t& m, bool& ok) {
                const int oy = oy0 + pr_o + 4 * j, ox = ox0 + pc_o;
                ok = oy < a.ho && ox < a.wo;
                m = ((int64_t)img * a.ho + oy) * a.wo + ox;
            };
            finish_wave_tile_lean<DT, 1, 2, false>(a, acc, cbase + ct * 32, hi, pix);
        }
    }
}

template <int DT>
static int launch_rw3(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int tiles_x = cdiv(a.wo, R2_T), tiles_y = cdiv(a.ho, R2_T);
    const int ntiles = a.n * tiles_x * tiles_y;
    const size_t lds = R2_BIAS_BYTES + (size_t)2 * R5_PATCH_BYTES;
    auto kfn = conv3x3_rw3_kernel<DT>;
    { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
    const int ny = a.cout / 128;
    int resident = 256 / ny;   // one 8-wave block per CU; the cout halves of a tile run side by side
    if (const char* e = getenv("YOLORT_AMD_RES3X3_BLOCKS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) resident = v;
    }
    a.nblk_m = ntiles;
    a.nblk_n = ny;
    hipLaunchKernelGGL(kfn, dim3(ntiles < resident ? ntiles : resident, ny), dim3(512), lds, s, a, tiles_x, tiles_y, ntiles);
    return check_launch("conv3x3_rw3_kernel");
}

// variant 1: cin = 64 -> cout = 128 (tile 134); variant 2: cin = 128 -> cout = 128 / 256, K split over two waves (tile 135)
int conv3x3_rw2_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s) {
    YMI_REQUIRE(variant >= 1 && variant <= 2, "ymi_conv2d: unknown stride-2 register-weights 3x3 variant %d", variant);
    if (variant == 2) {
        YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.ph == 1 && a.pw == 1 && a.sh == 2 && a.sw == 2 && a.cin == 128 && a.k_pad >= 9 * a.cin && (a.cout == 128 || a.cout == 256) &&
                        a.cout_pad >= a.cout && a.zeros != nullptr && a.up2 == 0 && a.split == 0 && a.chain_w == nullptr && a.res == nullptr && out_dtype == dtype && a.act == YMI_ACT_SILU,
                    "ymi_conv2d: the K-split stride-2 register-weights 3x3 kernel (tile 135) handles cin = 128, cout = 128 / 256, stride 2, pad 1, SiLU, 16-bit output, no shortcut / chained conv");
        YMI_REQUIRE(((int64_t)a.M + 1) * a.y_cs < ((int64_t)1 << 31) && (int64_t)a.n * a.h * a.w_in * a.x_cs < ((int64_t)1 << 31), "ymi_conv2d: tile 135: tensor too large for 32-bit offsets");
        return dtype == YMI_F16 ? launch_rw3<YMI_F16>(a, s) : launch_rw3<YMI_BF16>(a, s);
    }
    YMI_REQUIRE(a.kh == 3 && a.kw == 3 && a.ph == 1 && a.pw == 1 && a.sh == 2 && a.sw == 2 && a.cin == 64 && a.k_pad >= 9 * a.cin && a.cout == 128 && a.cout_pad >= a.cout &&
                    a.zeros != nullptr && a.up2 == 0 && a.split == 0 && a.chain_w == nullptr && a.res == nullptr && out_dtype == dtype && a.act == YMI_ACT_SILU,
                "ymi_conv2d: the stride-2 register-weights 3x3 kernel (tile 134) handles cin = 64, cout = 128, stride 2, pad 1, SiLU, 16-bit output, no shortcut / chained conv (and needs desc.zeros)");
    YMI_REQUIRE(((int64_t)a.M + 1) * a.y_cs < ((int64_t)1 << 31), "ymi_conv2d: tile 134: output tensor too large for 32-bit offsets");
    YMI_REQUIRE((int64_t)a.n * a.h * a.w_in * a.x_cs < ((int64_t)1 << 31), "ymi_conv2d: input tensor too large for 32-bit offsets");
    return dtype == YMI_F16 ? launch_rw2<YMI_F16>(a, s) : launch_rw2<YMI_BF16>(a, s);
}

}  // namespace ymi
