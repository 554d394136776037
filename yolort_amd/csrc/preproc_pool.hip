// HBM-bound helper kernels of the conv stack edges: letterbox, layout changes, SPP max-pool
// pyramid, nearest x2 upsample and channel-slice copy.  All move 16 bytes per lane where the
// layout allows (cdna_hip_programming.md G13) and are launched with >> 256 workgroups.
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace ymi {

// ---------------------------------------------------------------------------------------------
// Letterbox.  Replaces yolort/models/transform.py:53-97 (bilinear resize; ATen
// upsample_bilinear2d semantics: align_corners=False, scale recomputed as in/out, src clamped at
// 0, x1 = min(x0+1, in-1), fp32 lerp) and transform.py:297-330 (fill + centred copy).
// One thread per output pixel; up to LB_MAX images per launch (descriptors travel as kernargs so
// there is no device-side pointer table to allocate).
// ---------------------------------------------------------------------------------------------
constexpr int LB_MAX = 64;
struct LetterboxArgs {
    const void* img[LB_MAX];
    int geom[LB_MAX][6];  // h_in, w_in, h_res, w_res, pad_top, pad_left
    void* out;
    int n, hb, wb, c_out;
    float fill;
};

template <int IDT, int ODT>
__global__ __launch_bounds__(256) void letterbox_kernel(const LetterboxArgs a) {
    const int64_t npix = (int64_t)a.hb * a.wb;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix * a.n) return;
    const int img = (int)(gid / npix);
    const int rem = (int)(gid - (int64_t)img * npix);
    const int y = rem / a.wb, x = rem - y * a.wb;
    const int hin = a.geom[img][0], win = a.geom[img][1], hr = a.geom[img][2], wr = a.geom[img][3];
    const int yy = y - a.geom[img][4], xx = x - a.geom[img][5];
    float v[3] = {a.fill, a.fill, a.fill};
    if ((unsigned)yy < (unsigned)hr && (unsigned)xx < (unsigned)wr) {
        const float sy = (float)hin / (float)hr, sx = (float)win / (float)wr;
        float fy = __fsub_rn(__fmul_rn(sy, (float)yy + 0.5f), 0.5f);
        float fx = __fsub_rn(__fmul_rn(sx, (float)xx + 0.5f), 0.5f);
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        int y0 = (int)fy, x0 = (int)fx;
        y0 = y0 > hin - 1 ? hin - 1 : y0;
        x0 = x0 > win - 1 ? win - 1 : x0;
        const int y1 = y0 + 1 > hin - 1 ? hin - 1 : y0 + 1;
        const int x1 = x0 + 1 > win - 1 ? win - 1 : x0 + 1;
        float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        ly1 = ly1 < 0.f ? 0.f : (ly1 > 1.f ? 1.f : ly1);
        lx1 = lx1 < 0.f ? 0.f : (lx1 > 1.f ? 1.f : lx1);
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const int64_t plane = (int64_t)hin * win;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p00 = load_elem<IDT>(a.img[img], src_index<IDT>(c, y0, x0, win, plane));
            const float p01 = load_elem<IDT>(a.img[img], src_index<IDT>(c, y0, x1, win, plane));
            const float p10 = load_elem<IDT>(a.img[img], src_index<IDT>(c, y1, x0, win, plane));
            const float p11 = load_elem<IDT>(a.img[img], src_index<IDT>(c, y1, x1, win, plane));
            const float top = __fadd_rn(__fmul_rn(p00, lx0), __fmul_rn(p01, lx1));
            const float bot = __fadd_rn(__fmul_rn(p10, lx0), __fmul_rn(p11, lx1));
            v[c] = __fadd_rn(__fmul_rn(top, ly0), __fmul_rn(bot, ly1));
        }
    }
    if constexpr (ODT == YMI_F32) {
        float* o = (float*)a.out + gid * a.c_out;
        for (int c = 0; c < a.c_out; ++c) o[c] = c < 3 ? v[c] : 0.f;
    } else {
        uint16_t* o = (uint16_t*)a.out + gid * a.c_out;
        u32x2 w01;
        w01[0] = (uint32_t)to16<ODT>(v[0]) | ((uint32_t)to16<ODT>(v[1]) << 16);
        w01[1] = (uint32_t)to16<ODT>(v[2]);
        if (a.c_out == 4) {
            *reinterpret_cast<u32x2*>(o) = w01;
        } else if (a.c_out == 8) {
            u32x4 w = {w01[0], w01[1], 0u, 0u};
            *reinterpret_cast<u32x4*>(o) = w;
        } else {
            for (int c = 0; c < a.c_out; ++c) o[c] = c < 3 ? to16<ODT>(v[c]) : (uint16_t)0;
        }
    }
}

// Identity-size fast path (every image already has its resized size, the fixed-size stream case):
// the bilinear weights are exactly (1,0), so the result equals the source pixel bit for bit and the
// kernel degenerates into a planar-CHW -> NHWC4 interleave.  Four pixels per thread: one 8-byte load
// per plane (16-bit inputs), one 32-byte store.
template <int IDT, int ODT, int PX>   // PX = 4 or 8 pixels per thread
__global__ __launch_bounds__(256) void letterbox_copy_kernel(const LetterboxArgs a) {
    const int wq = a.wb / PX;
    const int64_t total = (int64_t)a.n * a.hb * wq;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int xq = (int)(gid % wq);
    const int64_t t = gid / wq;
    const int y = (int)(t % a.hb);
    const int img = (int)(t / a.hb);
    const int hin = a.geom[img][0], win = a.geom[img][1];
    const int yy = y - a.geom[img][4];
    const int x0 = xq * PX - a.geom[img][5];
    float v[PX][3];
    const bool row_in = (unsigned)yy < (unsigned)hin;
    const bool all_in = row_in && x0 >= 0 && x0 + PX - 1 < win && (win % PX == 0) && (a.geom[img][5] % PX == 0);
    const int64_t plane = (int64_t)hin * win;
    if (all_in && (IDT == YMI_F16 || IDT == YMI_BF16)) {
        constexpr int SDT = IDT == YMI_BF16 ? YMI_BF16 : YMI_F16;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint16_t* src = (const uint16_t*)a.img[img] + c * plane + (int64_t)yy * win + x0;
            uint32_t w[PX / 2];
            if constexpr (PX == 8) {
                const u32x4 p = *reinterpret_cast<const u32x4*>(src);
                w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; w[3] = p[3];
            } else {
                const u32x2 p = *reinterpret_cast<const u32x2*>(src);
                w[0] = p[0]; w[1] = p[1];
            }
#pragma unroll
            for (int i = 0; i < PX / 2; ++i) {
                v[2 * i][c] = from16<SDT>((uint16_t)(w[i] & 0xffff));
                v[2 * i + 1][c] = from16<SDT>((uint16_t)(w[i] >> 16));
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int xx = x0 + i;
            const bool in = row_in && (unsigned)xx < (unsigned)win;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[i][c] = in ? load_elem<IDT>(a.img[img], src_index<IDT>(c, yy, xx, win, plane)) : a.fill;
        }
    }
    if constexpr (ODT == YMI_F32) {
        float* o = (float*)a.out + (((int64_t)img * a.hb + y) * a.wb + xq * PX) * 4;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            f32x4 q = {v[i][0], v[i][1], v[i][2], 0.f};
            *reinterpret_cast<f32x4*>(o + 4 * i) = q;
        }
    } else {
        uint16_t* o = (uint16_t*)a.out + (((int64_t)img * a.hb + y) * a.wb + xq * PX) * 4;
#pragma unroll
        for (int i = 0; i < PX; i += 2) {   // two pixels (RGB0 RGB0) per 16-byte store
            u32x4 q;
            q[0] = (uint32_t)to16<ODT>(v[i][0]) | ((uint32_t)to16<ODT>(v[i][1]) << 16);
            q[1] = (uint32_t)to16<ODT>(v[i][2]);
            q[2] = (uint32_t)to16<ODT>(v[i + 1][0]) | ((uint32_t)to16<ODT>(v[i + 1][1]) << 16);
            q[3] = (uint32_t)to16<ODT>(v[i + 1][2]);
            *reinterpret_cast<u32x4*>(o + 8 * (i / 2)) = q;
        }
    }
}


// Tiled, LDS-staged form of the same letterbox (round 2): one thread per output pixel with twelve scalar 2-byte gathers ran at
// 0.23-0.26 of the HBM roofline -- the vector-memory pipe sees 12 load instructions per 8 output bytes.  Here a block owns
// LB_TH x LB_TW output pixels of one image: the source rows it needs are staged into LDS ONCE with aligned 16-byte loads
// (every source byte crosses the memory pipe once, in full 16-byte requests), the bilinear taps then come from LDS, and
// each thread writes TWO pixels = one 16-byte store (a wave covers 1 KiB of a canvas row).  Same arithmetic as
// letterbox_kernel, operation for operation: results are bit-identical.
constexpr int LB_TH = 4, LB_TW = 128;
constexpr int LB_RPW_DEFAULT = 4;
constexpr int LB_CG_DEFAULT = 1;    // 128-pixel column groups per tile of letterbox_tile2_kernel   // rows per wave of letterbox_tile2_kernel (tile = 4 * RPW rows x 128 columns)

struct LetterboxTileArgs {
    LetterboxArgs base;
    int tiles_x, tiles_y;
    int debug;   // tuning aid (YOLORT_AMD_LB_DEBUG): bit 0 = skip the staging loads, bit 1 = skip the resampling (fill only)
};

template <int IDT>
__device__ __forceinline__ float lds_elem(const unsigned char* p) {
    if constexpr (IDT == YMI_F16) { uint16_t v; __builtin_memcpy(&v, p, 2); return h2f(v); }
    else if constexpr (IDT == YMI_BF16) { uint16_t v; __builtin_memcpy(&v, p, 2); return bf2f(v); }
    else if constexpr (IDT == YMI_F32) { float v; __builtin_memcpy(&v, p, 4); return v; }
    else return (float)(*p) / 255.0f;
}


// Tiled letterbox (round 2, second pass).  The first tiled kernel (removed in round 5) issued ~560 vector instructions per thread for
// 16 output bytes (ISA count: ~100 per staged 16-byte chunk -- three runtime divisions in the chunk -> (plane, row, column)
// mapping and 64-bit address chains --, ~275 for the two pixels: the row alignment of each of the twelve taps recomputed with
// 64-bit multiplies).  Same data flow, same arithmetic, bit-identical results (tests/test_ops_gpu.py, every variant):
//   * a block owns (4 * RPW) x (128 * CG) output pixels; wave w produces rows w, w + 4, ... : the column taps (source offsets
//     and weights of the thread's two pixels) are computed once and reused for RPW rows, and the staged source rows are shared
//     by 4 * RPW output rows (fewer rows staged twice by vertically adjacent tiles);
//   * staging keeps the flat chunk order (every lane of every load busy) but decodes a chunk with one multiply-high by the
//     reciprocal of the row pitch (one division per block instead of three per chunk), four unconditional loads in flight
//     per thread, then four predicated LDS stores;
//   * the 16-byte alignment shift of a staged row is (A + plane * P + row * R) & 15 in 32-bit arithmetic;
//   * the taps are branch-free (a pixel outside the resized region reads offset 0 and keeps the fill value).
// Measured on the C3 batch (64 images, 8 shapes -> 1280x1280 bf16, 1.29 GB; tools/letterbox_bench.py, profiles/r02z_letterbox.txt):
// first tiled kernel 494 us; this kernel with row-per-wave staging (26 of 64 lanes busy per load on a 1.5x down-scale) 475 us --
// 2.3x fewer VALU instructions bought 4 %: the vector-memory pipe spends its address cycles per INSTRUCTION, not per lane;
// flat staging: RPW 2 -> 399 us, RPW 4 -> 374-380 us = 3.4 TB/s (0.43 of the 8 TB/s peak, 0.69 of a device-to-device copy of the
// canvas on the same box, 4.99 TB/s).  Final kernel with the staging loads removed 279 us, with the resampling removed 290 us,
// stores alone 178 us (YOLORT_AMD_LB_DEBUG = 1 / 2 / 3): loads and arithmetic each add ~100 us to the store stream of a block
// and do not overlap each other -- a block is load -> barrier -> compute -> store, overlap comes only from its neighbours.
template <int IDT, int ODT, int RPW, int CG>
__global__ __launch_bounds__(256) void letterbox_tile2_kernel(const LetterboxTileArgs t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lb_sm[];
    const LetterboxArgs& a = t.base;
    constexpr int TH = 4 * RPW;
    constexpr int ESZ = (IDT == YMI_F32) ? 4 : ((IDT == YMI_F16 || IDT == YMI_BF16) ? 2 : 1);
    constexpr bool HWC = IDT == YMI_U8_HWC;
    constexpr int BPP = HWC ? 3 : ESZ;
    constexpr int PLANES = HWC ? 1 : 3;
    constexpr int TW = LB_TW * CG;
    // (an XCD-aware tile order -- a contiguous run of tiles per XCD -- was measured: 0 .. -2 %, dropped)
    const int img = blockIdx.y, tile = blockIdx.x;
    const int ty = tile / t.tiles_x, tx = tile - ty * t.tiles_x;
    const int y0t = ty * TH, x0t = tx * TW;
    const int hin = a.geom[img][0], win = a.geom[img][1], hr = a.geom[img][2], wr = a.geom[img][3];
    const int pt = a.geom[img][4], pl = a.geom[img][5];
    const float sy = (float)hin / (float)hr, sx = (float)win / (float)wr;
    auto src = [](float s, int d, int n_in, int& i0, int& i1, float& l1) {   // exactly the arithmetic of letterbox_kernel
        float f = __fsub_rn(__fmul_rn(s, (float)d + 0.5f), 0.5f);
        f = f < 0.f ? 0.f : f;
        i0 = (int)f;
        i0 = i0 > n_in - 1 ? n_in - 1 : i0;
        i1 = i0 + 1 > n_in - 1 ? n_in - 1 : i0 + 1;
        l1 = f - (float)i0;
        l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    };
    const int ya = max(y0t, pt), yb = min(y0t + TH, min(pt + hr, a.hb));
    const int xa = max(x0t, pl), xb = min(x0t + TW, min(pl + wr, a.wb));
    const bool any = ya < yb && xa < xb;
    int ry0 = 0, ry1 = -1, cx0 = 0, cx1 = -1;
    if (any) {
        int i0, i1;
        float l;
        src(sy, ya - pt, hin, ry0, i1, l);
        src(sy, yb - 1 - pt, hin, i0, ry1, l);
        src(sx, xa - pl, win, cx0, i1, l);
        src(sx, xb - 1 - pl, win, i0, cx1, l);
    }
    const int nrows = ry1 - ry0 + 1;
    const int span = (cx1 - cx0 + 1) * BPP;
    const int pitch = ((span + 15 + 15) >> 4) << 4;
    const unsigned char* base = (const unsigned char*)a.img[img];
    const int64_t plane_b = (int64_t)hin * win * ESZ;
    const int64_t row_b = (int64_t)win * BPP;
    // 16-byte alignment of the first needed byte of (plane c, source row r): (al_a + c * al_p + r * al_r) & 15
    const int al_a = (int)(((uintptr_t)base + (uintptr_t)((int64_t)cx0 * BPP)) & 15);
    const int al_p = (int)(plane_b & 15), al_r = (int)(row_b & 15);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (any && !(t.debug & 1)) {
        // Flat chunk order: chunk i of the tile is 16-byte column (i % cpr) of staged row (i / cpr) and lands at LDS byte 16 * i
        // (pitch = 16 * cpr), so every load / LDS-store instruction has all 64 lanes busy -- the vector-memory pipe spends its
        // 16 address cycles per instruction whatever the number of active lanes, and a row-per-wave mapping (26 of 64 lanes
        // on a 1.5x down-scale) measured 1.35x slower.  One division per block (the reciprocal of cpr), none per chunk.
        const unsigned cpr = (unsigned)pitch >> 4;
        const unsigned total = (unsigned)(PLANES * nrows) * cpr;
        // i / cpr == mulhi(i, magic) for i < 2^16 (total <= 60 KiB / 16).  cpr == 1 (a tile that overlaps the resized region in columns that all map to ONE
        // uint8 source column: span 1, pitch 16) would wrap the magic to 0 and stage plane 0 / row 0 only: there the chunk IS the job
        const unsigned magic = cpr == 1 ? 0u : 0xffffffffu / cpr + 1u;
        const unsigned char* gb = base + (int64_t)cx0 * BPP;
        const unsigned char* safe = (const unsigned char*)((uintptr_t)base & ~(uintptr_t)15);   // a chunk that is not needed reads this one instead (never stored)
        for (unsigned i0 = threadIdx.x; i0 < total; i0 += 1024) {
            u32x4 buf[4];
            bool have[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // four unconditional loads in flight per thread, then four predicated LDS stores
                const unsigned i = i0 + 256u * u;
                const unsigned job = cpr == 1 ? i : __umulhi(i, magic), k = i - job * cpr;
                const int c = PLANES == 1 ? 0 : ((int)job >= 2 * nrows ? 2 : ((int)job >= nrows ? 1 : 0));
                const int r = ry0 + (int)job - c * nrows;
                const int sh = (al_a + c * al_p + r * al_r) & 15;
                have[u] = i < total && (int)(k << 4) < sh + span;   // aligned 16-byte chunks holding at least one needed byte (an aligned chunk never crosses a page)
                const unsigned char* g = gb + c * plane_b + (int64_t)r * row_b + (int64_t)((int)(k << 4) - sh);
                buf[u] = *reinterpret_cast<const u32x4*>(have[u] ? g : safe);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (have[u]) *reinterpret_cast<u32x4*>(lb_sm + (size_t)(i0 + 256u * u) * 16) = buf[u];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int cg = 0; cg < CG; ++cg) {
    const int x = x0t + cg * LB_TW + (threadIdx.x & 63) * 2;
    if (x >= a.wb) break;
    // column taps of the thread's two pixels: byte offsets inside a staged row, weights
    bool inx[2];
    int ox0[2], ox1[2];
    float lx0[2], lx1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int xx = x + p - pl;
        inx[p] = (unsigned)xx < (unsigned)wr && x + p < a.wb;
        ox0[p] = ox1[p] = 0;
        lx0[p] = lx1[p] = 0.f;
        if (inx[p]) {
            int sx0, sx1;
            src(sx, xx, win, sx0, sx1, lx1[p]);
            lx0[p] = 1.f - lx1[p];
            ox0[p] = (sx0 - cx0) * BPP;
            ox1[p] = (sx1 - cx0) * BPP;
        }
    }
    const bool two = x + 1 < a.wb;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = y0t + 4 * j + wv;
        if (y >= a.hb) break;   // wave-uniform
        float v[2][3];
#pragma unroll
        for (int p = 0; p < 2; ++p) v[p][0] = v[p][1] = v[p][2] = a.fill;
        const int yy = y - pt;
        if ((unsigned)yy < (unsigned)hr && (inx[0] || inx[1]) && !(t.debug & 2)) {
            int sy0, sy1;
            float ly1;
            src(sy, yy, hin, sy0, sy1, ly1);
            const float ly0 = 1.f - ly1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int pc = HWC ? 0 : c;
                const int sub = HWC ? c : 0;
                const unsigned char* r0 = lb_sm + (pc * nrows + (sy0 - ry0)) * pitch + ((al_a + pc * al_p + sy0 * al_r) & 15) + sub;
                const unsigned char* r1 = lb_sm + (pc * nrows + (sy1 - ry0)) * pitch + ((al_a + pc * al_p + sy1 * al_r) & 15) + sub;
#pragma unroll
                for (int p = 0; p < 2; ++p) {   // branch-free: a pixel outside the resized region reads offset 0 of the rows and keeps the fill value
                    const float p00 = lds_elem<IDT>(r0 + ox0[p]), p01 = lds_elem<IDT>(r0 + ox1[p]);
                    const float p10 = lds_elem<IDT>(r1 + ox0[p]), p11 = lds_elem<IDT>(r1 + ox1[p]);
                    const float top = __fadd_rn(__fmul_rn(p00, lx0[p]), __fmul_rn(p01, lx1[p]));
                    const float bot = __fadd_rn(__fmul_rn(p10, lx0[p]), __fmul_rn(p11, lx1[p]));
                    const float val = __fadd_rn(__fmul_rn(top, ly0), __fmul_rn(bot, ly1));
                    v[p][c] = inx[p] ? val : a.fill;
                }
            }
        }
        const int64_t o = (((int64_t)img * a.hb + y) * a.wb + x) * 4;
        if constexpr (ODT == YMI_F32) {
            float* op = (float*)a.out + o;
            f32x4 q0 = {v[0][0], v[0][1], v[0][2], 0.f};
            *reinterpret_cast<f32x4*>(op) = q0;
            if (two) {
                f32x4 q1 = {v[1][0], v[1][1], v[1][2], 0.f};
                *reinterpret_cast<f32x4*>(op + 4) = q1;
            }
        } else {
            uint16_t* op = (uint16_t*)a.out + o;
            u32x4 q;
            q[0] = cvt_pk16<ODT>(f32x2{v[0][0], v[0][1]});   // hardware pair conversions (round 4; same rounding as to16: bf16 output was ~40 VALU per pixel pair in software)
            q[1] = cvt_pk16<ODT>(f32x2{v[0][2], 0.f});
            q[2] = cvt_pk16<ODT>(f32x2{v[1][0], v[1][1]});
            q[3] = cvt_pk16<ODT>(f32x2{v[1][2], 0.f});
            if (two && (a.wb & 1) == 0) *reinterpret_cast<u32x4*>(op) = q;
            else {
                u32x2 h0 = {q[0], q[1]};
                *reinterpret_cast<u32x2*>(op) = h0;
                if (two) { u32x2 h1 = {q[2], q[3]}; *reinterpret_cast<u32x2*>(op + 4) = h1; }
            }
        }
    }
    }   // column groups
}

// (Round 4 also built DMA-staged forms of this kernel -- every chunk of a tile in flight at once, and a persistent double-buffered one with counted vmcnt: 355 / 460 us
// against 360 us on the C3 batch, profiles/r04u_letterbox_variants.txt.  The kernel is instruction-issue-bound, not load-bound; they were removed in round 5.)

template <int IDT>
static size_t letterbox_tile_lds(const LetterboxArgs& a, int th = LB_TH, int tw = LB_TW, size_t cap = 64 * 1024) {
    constexpr int ESZ = (IDT == YMI_F32) ? 4 : ((IDT == YMI_F16 || IDT == YMI_BF16) ? 2 : 1);
    constexpr int BPP = IDT == YMI_U8_HWC ? 3 : ESZ;
    constexpr int PLANES = IDT == YMI_U8_HWC ? 1 : 3;
    size_t need = 16;
    for (int i = 0; i < a.n; ++i) {
        const double sy = (double)a.geom[i][0] / a.geom[i][2], sx = (double)a.geom[i][1] / a.geom[i][3];
        const int rows = (int)(th * sy) + 3, cols = (int)(tw * sx) + 3;
        const size_t pitch = (((size_t)(cols < a.geom[i][1] ? cols : a.geom[i][1]) * BPP + 30) >> 4) << 4;
        const size_t b = (size_t)PLANES * (rows < a.geom[i][0] ? rows : a.geom[i][0]) * pitch;
        need = b > need ? b : need;
    }
    return need <= cap ? need : 0;   // default cap: the largest dynamic LDS request that needs no opt-in
}

template <int IDT>
static int letterbox_dispatch(const LetterboxArgs& a, int out_dtype, hipStream_t s) {
    bool identity = a.c_out == 4 && a.wb % 4 == 0;
    for (int i = 0; i < a.n && identity; ++i) identity = a.geom[i][0] == a.geom[i][2] && a.geom[i][1] == a.geom[i][3];
    if (identity) {   // no resampling anywhere in this launch: interleave-copy kernel (bit-identical results)
        const int px = a.wb % 8 == 0 ? 8 : 4;
        const int64_t tq = (int64_t)a.n * a.hb * (a.wb / px);
        dim3 gq((unsigned)((tq + 255) / 256)), bq(256);
#define YMI_LBC(ODT_)                                                                             \
    if (px == 8) hipLaunchKernelGGL((letterbox_copy_kernel<IDT, ODT_, 8>), gq, bq, 0, s, a);      \
    else hipLaunchKernelGGL((letterbox_copy_kernel<IDT, ODT_, 4>), gq, bq, 0, s, a);
        switch (out_dtype) {
            case YMI_F16: YMI_LBC(YMI_F16) break;
            case YMI_BF16: YMI_LBC(YMI_BF16) break;
            case YMI_F32: YMI_LBC(YMI_F32) break;
            default: set_error("ymi_letterbox: bad out_dtype %d", out_dtype); return YMI_EINVAL;
        }
#undef YMI_LBC
        return check_launch("letterbox_copy_kernel");
    }
    // tuning aid: YOLORT_AMD_LETTERBOX = "pixel" (per-pixel kernel), "1" / "2" / "4" (rows per wave of the tiled kernel; default: the largest
    // of 4, 2, 1 whose staged rows fit the LDS budget)
    const char* lb_env = getenv("YOLORT_AMD_LETTERBOX");   // read per call (one launch per batch): tests switch kernels in-process
    const bool lb_pixel = lb_env && !strcmp(lb_env, "pixel");
    const int rpw_max = (lb_env && (lb_env[0] == '1' || lb_env[0] == '2' || lb_env[0] == '4') && lb_env[1] == 0) ? lb_env[0] - '0' : LB_RPW_DEFAULT;
    if (a.c_out == 4 && !lb_pixel) {
        const char* dbg = getenv("YOLORT_AMD_LB_DEBUG");
        const char* cge = getenv("YOLORT_AMD_LB_CG");
        const int cg_max = cge ? (atoi(cge) >= 2 ? 2 : 1) : LB_CG_DEFAULT;
        for (int rpw = rpw_max; rpw >= 1; rpw >>= 1) {
            int cg = cg_max;
            size_t lds = letterbox_tile_lds<IDT>(a, 4 * rpw, LB_TW * cg);
            if (lds == 0 && cg > 1) lds = letterbox_tile_lds<IDT>(a, 4 * rpw, LB_TW * (cg = 1));
            if (lds == 0) continue;
            LetterboxTileArgs t;
            t.base = a;
            t.tiles_x = cdiv(a.wb, LB_TW * cg);
            t.tiles_y = cdiv(a.hb, 4 * rpw);
            t.debug = dbg ? atoi(dbg) : 0;
            dim3 gt((unsigned)(t.tiles_x * t.tiles_y), (unsigned)a.n), bt(256);
#define YMI_LBT2C(ODT_, CG_)                                                                                        \
    if (rpw == 4) hipLaunchKernelGGL((letterbox_tile2_kernel<IDT, ODT_, 4, CG_>), gt, bt, lds, s, t);                \
    else if (rpw == 2) hipLaunchKernelGGL((letterbox_tile2_kernel<IDT, ODT_, 2, CG_>), gt, bt, lds, s, t);           \
    else hipLaunchKernelGGL((letterbox_tile2_kernel<IDT, ODT_, 1, CG_>), gt, bt, lds, s, t);
#define YMI_LBT2(ODT_)                                  \
    if (cg == 2) { YMI_LBT2C(ODT_, 2) } else { YMI_LBT2C(ODT_, 1) }
            switch (out_dtype) {
                case YMI_F16: YMI_LBT2(YMI_F16) break;
                case YMI_BF16: YMI_LBT2(YMI_BF16) break;
                case YMI_F32: YMI_LBT2(YMI_F32) break;
                default: set_error("ymi_letterbox: bad out_dtype %d", out_dtype); return YMI_EINVAL;
            }
#undef YMI_LBT2
#undef YMI_LBT2C
            return check_launch("letterbox_tile2_kernel");
        }
    }
    const int64_t total = (int64_t)a.n * a.hb * a.wb;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    switch (out_dtype) {
        case YMI_F16: hipLaunchKernelGGL((letterbox_kernel<IDT, YMI_F16>), grid, block, 0, s, a); break;
        case YMI_BF16: hipLaunchKernelGGL((letterbox_kernel<IDT, YMI_BF16>), grid, block, 0, s, a); break;
        case YMI_F32: hipLaunchKernelGGL((letterbox_kernel<IDT, YMI_F32>), grid, block, 0, s, a); break;
        default: set_error("ymi_letterbox: bad out_dtype %d", out_dtype); return YMI_EINVAL;
    }
    return check_launch("letterbox_kernel");
}

// ---------------------------------------------------------------------------------------------
// NCHW <-> NHWC (module-level API edges only)
// ---------------------------------------------------------------------------------------------
template <int IDT, int ODT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const void* x, int n, int c, int h, int w, void* y, int y_cs, int c_pad) {
    const int64_t npix = (int64_t)n * h * w;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix) return;
    const int64_t hw = (int64_t)h * w;
    const int img = (int)(gid / hw);
    const int64_t p = gid - img * hw;
    for (int ch = 0; ch < c_pad; ++ch) {
        const float v = ch < c ? load_elem<IDT>(x, ((int64_t)img * c + ch) * hw + p) : 0.f;
        if constexpr (ODT == YMI_F32) ((float*)y)[gid * y_cs + ch] = v;
        else ((uint16_t*)y)[gid * y_cs + ch] = to16<ODT>(v);
    }
}
template <int IDT, int ODT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* x, int x_cs, int n, int c, int h, int w, void* y) {
    const int64_t total = (int64_t)n * c * h * w;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int64_t hw = (int64_t)h * w;
    const int64_t p = gid % hw;
    const int64_t t = gid / hw;
    const int ch = (int)(t % c);
    const int img = (int)(t / c);
    const float v = load_elem<IDT>(x, ((int64_t)img * hw + p) * x_cs + ch);
    if constexpr (ODT == YMI_F32) ((float*)y)[gid] = v;
    else ((uint16_t*)y)[gid] = to16<ODT>(v);
}

// ---------------------------------------------------------------------------------------------
// SPP pyramid: pool5/9/13 (stride 1, same, -inf pad) of channels [0,c) -> slices 1..3.
// max is exact, so one pass over the 13x13 window carrying three running maxima equals the
// reference's three MaxPool2d calls (common.py:183) bit for bit.  8 channels (16 B) per thread.
// ---------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void spp_pool_kernel(uint16_t* buf, int n, int h, int w, int c, int cs) {
    const int c8 = c / 8;
    const int64_t total = (int64_t)n * h * w * c8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cc = (int)(gid % c8) * 8;
    const int64_t pix = gid / c8;
    const int x = (int)(pix % w);
    const int y = (int)((pix / w) % h);
    const int img = (int)(pix / ((int64_t)w * h));
    float m5[8], m9[8], m13[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m5[e] = m9[e] = m13[e] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)h) continue;
        const int ady = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)w) continue;
            const int adx = dx < 0 ? -dx : dx;
            const int r = ady > adx ? ady : adx;
            const u32x4 v = *reinterpret_cast<const u32x4*>(buf + ((int64_t)(img * h + yy) * w + xx) * cs + cc);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = from16<DT>((uint16_t)((v[e >> 1] >> ((e & 1) * 16)) & 0xffff));
                m13[e] = fmaxf(m13[e], f);
                if (r <= 4) m9[e] = fmaxf(m9[e], f);
                if (r <= 2) m5[e] = fmaxf(m5[e], f);
            }
        }
    }
    uint16_t* o = buf + pix * cs + cc;
    u32x4 o5, o9, o13;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o5[e] = (uint32_t)to16<DT>(m5[2 * e]) | ((uint32_t)to16<DT>(m5[2 * e + 1]) << 16);
        o9[e] = (uint32_t)to16<DT>(m9[2 * e]) | ((uint32_t)to16<DT>(m9[2 * e + 1]) << 16);
        o13[e] = (uint32_t)to16<DT>(m13[2 * e]) | ((uint32_t)to16<DT>(m13[2 * e + 1]) << 16);
    }
    *reinterpret_cast<u32x4*>(o + c) = o5;
    *reinterpret_cast<u32x4*>(o + 2 * c) = o9;
    *reinterpret_cast<u32x4*>(o + 3 * c) = o13;
}

// fp32 parity-mode form of the pyramid (4 channels = 16 B per thread, direct 13x13 window; max is exact)
__global__ __launch_bounds__(256) void spp_pool_f32_kernel(float* buf, int n, int h, int w, int c, int cs) {
    const int c4 = c / 4;
    const int64_t total = (int64_t)n * h * w * c4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cc = (int)(gid % c4) * 4;
    const int64_t pix = gid / c4;
    const int x = (int)(pix % w);
    const int y = (int)((pix / w) % h);
    const int img = (int)(pix / ((int64_t)w * h));
    f32x4 m5, m9, m13;
#pragma unroll
    for (int e = 0; e < 4; ++e) m5[e] = m9[e] = m13[e] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)h) continue;
        const int ady = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)w) continue;
            const int adx = dx < 0 ? -dx : dx;
            const int r = ady > adx ? ady : adx;
            const f32x4 v = *reinterpret_cast<const f32x4*>(buf + ((int64_t)(img * h + yy) * w + xx) * cs + cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m13[e] = fmaxf(m13[e], v[e]);
                if (r <= 4) m9[e] = fmaxf(m9[e], v[e]);
                if (r <= 2) m5[e] = fmaxf(m5[e], v[e]);
            }
        }
    }
    float* o = buf + pix * cs + cc;
    *reinterpret_cast<f32x4*>(o + c) = m5;
    *reinterpret_cast<f32x4*>(o + 2 * c) = m9;
    *reinterpret_cast<f32x4*>(o + 3 * c) = m13;
}

// LDS cascade form of the same pyramid: one block per (image, CPB-channel chunk), CPB = 32 (8 when c % 32 != 0).
// The h x w plane of those channels is staged in LDS in its storage type and three 5x5 max stages are applied back to
// back, each separable (row pass then column pass): mp9 = mp5(mp5(x)), mp13 = mp5(mp9) -- the SPPF identity of the
// reference (common.py:196), exact because max is exact (also in fp16 / bf16: no arithmetic, only selection).
// 10 taps per stage instead of a 169-tap window, the plane is read from HBM once, and with CPB = 32 every pixel
// moves as one 64-byte run (the 8-channel form touched 16 bytes of each 2 KiB pixel row per block).
template <int DT>
__device__ __forceinline__ u32x4 max8(const u32x4& p, const u32x4& q) {
    u32x4 r;
    if constexpr (DT == YMI_F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t pe = p[e], qe = q[e];
            h2 x, y;
            __builtin_memcpy(&x, &pe, 4);
            __builtin_memcpy(&y, &qe, 4);
            const h2 m = __builtin_elementwise_max(x, y);   // v_pk_max_f16
            uint32_t me;
            __builtin_memcpy(&me, &m, 4);
            r[e] = me;
        }
    } else {
        // bfloat16 has no packed max: the plane is held in LDS as ORDER KEYS (spp_key16: sign-magnitude -> unsigned order, two per dword) and the maximum of two
        // keys is `v_pk_max_u16` -- one instruction per dword like the fp16 form.  (Round 3 converted both operands to fp32, compared and re-selected per element:
        // ~14 VALU per dword, 1 300 per 8-channel group over the three stages -- yolov5m's 40 x 40 pool was bound by exactly that: 300 us for 315 MB.)
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t pe = p[e], qe = q[e];
            us2 x, y;
            __builtin_memcpy(&x, &pe, 4);
            __builtin_memcpy(&y, &qe, 4);
            const us2 m = __builtin_elementwise_max(x, y);   // v_pk_max_u16
            uint32_t me;
            __builtin_memcpy(&me, &m, 4);
            r[e] = me;
        }
    }
    return r;
}
// bfloat16 <-> order key, two values per dword: negative values have all bits flipped, non-negative ones the sign bit set -- unsigned comparison of the keys is the
// numerical order of the values (-0 below +0; the pool never sees a NaN).  fp16 planes stay as they are (DT-dispatched no-ops).
template <int DT>
__device__ __forceinline__ u32x4 spp_key16(const u32x4& v) {
    if constexpr (DT == YMI_F16) return v;
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t neg = ((v[e] >> 15) & 0x00010001u) * 0xffffu;   // 0xffff in every half whose sign bit is set
        r[e] = v[e] ^ (neg | 0x80008000u);
    }
    return r;
}
template <int DT>
__device__ __forceinline__ u32x4 spp_unkey16(const u32x4& k) {
    if constexpr (DT == YMI_F16) return k;
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t pos = ((k[e] >> 15) & 0x00010001u) * 0xffffu;   // keys of non-negative values have the top bit set
        r[e] = k[e] ^ ((~pos) | 0x80008000u);
    }
    return r;
}

// Block size (round 4): 256 ... 1024 threads.  A block walks its plane in steps of blockDim.x / G pixels through eight phases (load, 3 x row pass, 3 x column pass + store)
// separated by barriers, and the LDS footprint allows one to three blocks per CU: with 256 threads a CU had 4-12 waves in a chain of dependent LDS round trips
// (yolov5s 20 x 20: 21 us for 26 MB; yolov5m 40 x 40: 387 us at 0.10 of the HBM bound).  The threads only divide the pixels among themselves: results do not depend on the block size.
template <int DT, int G>   // G = 16-byte groups (8 channels) per pixel handled by one block
__global__ __launch_bounds__(1024) void spp_pool_lds_kernel(uint16_t* buf, int h, int w, int c, int cs) {
    extern __shared__ __attribute__((aligned(16))) u32x4 spp_sm[];
    const int hw = h * w;
    u32x4* A = spp_sm;                 // [hw][G]: the current stage's plane (x, then mp5, mp9 -- the column pass overwrites it in place: the row pass was its last reader)
    u32x4* B = spp_sm + (size_t)hw * G;   // row-pass scratch
    const int cg = c / (8 * G);
    const int img = blockIdx.x / cg, cc = (blockIdx.x % cg) * 8 * G;
    uint16_t* base = buf + (int64_t)img * hw * cs + cc;
    const int sub = threadIdx.x % G;              // channel group of this thread
    const int p0 = threadIdx.x / G;               // first pixel
    const int PSTEP = (int)blockDim.x / G;
    for (int p = p0; p < hw; p += PSTEP) A[p * G + sub] = spp_key16<DT>(*reinterpret_cast<const u32x4*>(base + (int64_t)p * cs + sub * 8));
    __syncthreads();
    u32x4* src = A;
    u32x4* dst = A;
    for (int stage = 0; stage < 3; ++stage) {
        for (int p = p0; p < hw; p += PSTEP) {   // row pass: src -> B
            const int y = p / w, x = p - y * w;
            const int x0 = x - 2 < 0 ? 0 : x - 2, x1 = x + 2 > w - 1 ? w - 1 : x + 2;
            u32x4 m = src[(y * w + x0) * G + sub];
            for (int xx = x0 + 1; xx <= x1; ++xx) m = max8<DT>(m, src[(y * w + xx) * G + sub]);
            B[p * G + sub] = m;
        }
        __syncthreads();
        for (int p = p0; p < hw; p += PSTEP) {   // column pass: B -> dst, and out to HBM
            const int y = p / w, x = p - y * w;
            const int y0 = y - 2 < 0 ? 0 : y - 2, y1 = y + 2 > h - 1 ? h - 1 : y + 2;
            u32x4 m = B[(y0 * w + x) * G + sub];
            for (int yy = y0 + 1; yy <= y1; ++yy) m = max8<DT>(m, B[(yy * w + x) * G + sub]);
            *reinterpret_cast<u32x4*>(base + (int64_t)p * cs + (stage + 1) * c + sub * 8) = spp_unkey16<DT>(m);
            dst[p * G + sub] = m;
        }
        __syncthreads();
    }
}

// nearest x2 upsample: one thread per (input pixel, 8 channels) -> 4 output pixels
__global__ __launch_bounds__(256) void upsample2x_kernel(const uint16_t* x, int x_cs, int n, int h, int w, int c, uint16_t* y, int y_cs) {
    const int c8 = c / 8;
    const int64_t total = (int64_t)n * h * w * c8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cc = (int)(gid % c8) * 8;
    const int64_t pix = gid / c8;
    const int xx = (int)(pix % w);
    const int yy = (int)((pix / w) % h);
    const int img = (int)(pix / ((int64_t)w * h));
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + pix * x_cs + cc);
    const int w2 = 2 * w;
    uint16_t* o = y + ((int64_t)(img * 2 * h + 2 * yy) * w2 + 2 * xx) * y_cs + cc;
    *reinterpret_cast<u32x4*>(o) = v;
    *reinterpret_cast<u32x4*>(o + y_cs) = v;
    *reinterpret_cast<u32x4*>(o + (int64_t)w2 * y_cs) = v;
    *reinterpret_cast<u32x4*>(o + (int64_t)w2 * y_cs + y_cs) = v;
}

__global__ __launch_bounds__(256) void copy_view_kernel(const uint16_t* x, int x_cs, int64_t npix, int c, uint16_t* y, int y_cs) {
    const int c8 = c / 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix * c8) return;
    const int cc = (int)(gid % c8) * 8;
    const int64_t pix = gid / c8;
    *reinterpret_cast<u32x4*>(y + pix * y_cs + cc) = *reinterpret_cast<const u32x4*>(x + pix * x_cs + cc);
}

// ---- activations of the legacy r3.1 blocks as a launch of their own (round 5) ------------------------------------------------------------------------------------
// Hardswish (common.py:64-65, `Conv(version="r3.1")`) and LeakyReLU(0.1) (common.py:140, `BottleneckCSP.act`), with the Bottleneck's shortcut (common.py:115-116) added
// AFTER the activation like the reference does.  The convolution epilogues carry SiLU / identity only: folding these two into the general epilogue of every conv kernel
// moved the register allocation of kernels that sat at their limit (the 256-wide 8-wave implicit GEMM: 2-3 x slower on the r6.0 plans, profiles/r05k_*), so an r3.1
// convolution runs with YMI_ACT_NONE and this kernel rewrites its output in place: one 16-byte packet (8 halves / 4 floats) per thread, HBM-bound.
// fp32 mode: torch's own order of operations -- (x * min(max(x + 3, 0), 6)) / 6 with a true division; x > 0 ? x : 0.1 x -- on the exactly stored fp32 pre-activation.
__device__ __forceinline__ float act_legacy_f32(float v, int act) {
    if (act == YMI_ACT_HARDSWISH) return (v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f)) / 6.0f;
    return v > 0.0f ? v : v * 0.1f;
}

template <int DT>
__global__ __launch_bounds__(256) void act_kernel(uint16_t* y, int y_cs, int64_t npix, int c, int act, const uint16_t* res, int res_cs) {
    const int c8 = c / 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix * c8) return;
    const int cc = (int)(gid % c8) * 8;
    const int64_t pix = gid / c8;
    u32x4 v = *reinterpret_cast<const u32x4*>(y + pix * y_cs + cc);
    u32x4 r = {0u, 0u, 0u, 0u};
    if (res != nullptr) r = *reinterpret_cast<const u32x4*>(res + pix * res_cs + cc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float lo = from16<DT>((uint16_t)(v[q] & 0xffff)), hi = from16<DT>((uint16_t)(v[q] >> 16));
        lo = act_legacy_f32(lo, act);
        hi = act_legacy_f32(hi, act);
        if (res != nullptr) {
            lo += from16<DT>((uint16_t)(r[q] & 0xffff));
            hi += from16<DT>((uint16_t)(r[q] >> 16));
        }
        v[q] = (uint32_t)to16<DT>(lo) | ((uint32_t)to16<DT>(hi) << 16);
    }
    *reinterpret_cast<u32x4*>(y + pix * y_cs + cc) = v;
}

__global__ __launch_bounds__(256) void act_f32_kernel(float* y, int y_cs, int64_t npix, int c, int act, const float* res, int res_cs) {
    const int c4 = c / 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix * c4) return;
    const int cc = (int)(gid % c4) * 4;
    const int64_t pix = gid / c4;
    f32x4 v = *reinterpret_cast<const f32x4*>(y + pix * y_cs + cc);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = act_legacy_f32(v[q], act);
    if (res != nullptr) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(res + pix * res_cs + cc);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += r[q];
    }
    *reinterpret_cast<f32x4*>(y + pix * y_cs + cc) = v;
}

// fp32 mode: the SPPF cascade in LDS (round 5; the direct 169-tap kernel above took 155 us on the yolov5s bs-32 plan against 17 us for the 16-bit LDS kernel).
// A block owns the whole h x w plane of ONE image and FOUR channels (16 bytes per pixel): plane -> LDS, then three times {5-wide row maximum into the scratch plane,
// 5-tall column maximum back into the plane, store as the next concat slot}.  max is exact and mp5(mp5(x)) = mp9(x), mp5(mp9(x)) = mp13(x) (common.py:196), so the three
// slots equal the direct windows bit for bit (-inf padding = the clipped window).
__global__ __launch_bounds__(1024) void spp_pool_f32_lds_kernel(float* buf, int h, int w, int c, int cs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spp32_sm[];
    f32x4* cur = reinterpret_cast<f32x4*>(spp32_sm);
    f32x4* tmp = cur + h * w;
    const int groups = c / 4;
    const int img = blockIdx.x / groups, cg = blockIdx.x - img * groups;
    float* base = buf + (int64_t)img * h * w * cs + cg * 4;
    const int npix = h * w;
    for (int p = threadIdx.x; p < npix; p += blockDim.x) cur[p] = *reinterpret_cast<const f32x4*>(base + (int64_t)p * cs);
    __syncthreads();
    for (int stage = 1; stage <= 3; ++stage) {
        for (int p = threadIdx.x; p < npix; p += blockDim.x) {   // rows
            const int y = p / w, x = p - y * w;
            const int x0 = x - 2 < 0 ? 0 : x - 2, x1 = x + 2 > w - 1 ? w - 1 : x + 2;
            f32x4 m = cur[y * w + x0];
            for (int xx = x0 + 1; xx <= x1; ++xx) {
                const f32x4 v = cur[y * w + xx];
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            tmp[p] = m;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < npix; p += blockDim.x) {   // columns
            const int y = p / w, x = p - y * w;
            const int y0 = y - 2 < 0 ? 0 : y - 2, y1 = y + 2 > h - 1 ? h - 1 : y + 2;
            f32x4 m = tmp[y0 * w + x];
            for (int yy = y0 + 1; yy <= y1; ++yy) {
                const f32x4 v = tmp[yy * w + x];
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            cur[p] = m;
            *reinterpret_cast<f32x4*>(base + (int64_t)p * cs + stage * c) = m;
        }
        __syncthreads();
    }
}

}  // namespace ymi

using namespace ymi;

extern "C" int ymi_letterbox(const void* const* imgs, const int32_t* geom, int n, int in_dtype, void* out, int hb, int wb,
                             int c_out, int out_dtype, float fill, void* stream) {
    YMI_REQUIRE(imgs && geom && out && n >= 0, "ymi_letterbox: null argument");
    YMI_REQUIRE(c_out >= 3, "ymi_letterbox: c_out must be >= 3");
    const size_t esz = out_dtype == YMI_F32 ? 4 : 2;
    for (int base = 0; base < n; base += LB_MAX) {
        LetterboxArgs a;
        a.n = (n - base) < LB_MAX ? (n - base) : LB_MAX;
        for (int i = 0; i < a.n; ++i) {
            a.img[i] = imgs[base + i];
            for (int k = 0; k < 6; ++k) a.geom[i][k] = geom[(base + i) * 6 + k];
            YMI_REQUIRE(a.geom[i][2] + a.geom[i][4] <= hb && a.geom[i][3] + a.geom[i][5] <= wb && a.geom[i][4] >= 0 && a.geom[i][5] >= 0,
                        "ymi_letterbox: image %d (%dx%d at %d,%d) does not fit the %dx%d canvas", base + i, a.geom[i][2], a.geom[i][3], a.geom[i][4], a.geom[i][5], hb, wb);
        }
        a.out = (char*)out + (size_t)base * hb * wb * c_out * esz;
        a.hb = hb; a.wb = wb; a.c_out = c_out; a.fill = fill;
        int rc;
        switch (in_dtype) {
            case YMI_F32: rc = letterbox_dispatch<YMI_F32>(a, out_dtype, (hipStream_t)stream); break;
            case YMI_F16: rc = letterbox_dispatch<YMI_F16>(a, out_dtype, (hipStream_t)stream); break;
            case YMI_BF16: rc = letterbox_dispatch<YMI_BF16>(a, out_dtype, (hipStream_t)stream); break;
            case YMI_U8: rc = letterbox_dispatch<YMI_U8>(a, out_dtype, (hipStream_t)stream); break;
            case YMI_U8_HWC: rc = letterbox_dispatch<YMI_U8_HWC>(a, out_dtype, (hipStream_t)stream); break;
            default: set_error("ymi_letterbox: bad in_dtype %d", in_dtype); return YMI_EINVAL;
        }
        if (rc != YMI_OK) return rc;
    }
    return YMI_OK;
}

#define YMI_DISPATCH_IO(KERNEL, IDT_V, ODT_V, ...)                                                          \
    do {                                                                                                    \
        bool _done = false;                                                                                 \
        auto _try = [&](auto idt, auto odt) {                                                               \
            if (!_done && IDT_V == decltype(idt)::value && ODT_V == decltype(odt)::value) {                 \
                hipLaunchKernelGGL((KERNEL<decltype(idt)::value, decltype(odt)::value>), grid, block, 0, s, __VA_ARGS__); \
                _done = true;                                                                               \
            }                                                                                               \
        };                                                                                                  \
        _try(std::integral_constant<int, YMI_F32>{}, std::integral_constant<int, YMI_F16>{});               \
        _try(std::integral_constant<int, YMI_F32>{}, std::integral_constant<int, YMI_BF16>{});              \
        _try(std::integral_constant<int, YMI_F32>{}, std::integral_constant<int, YMI_F32>{});               \
        _try(std::integral_constant<int, YMI_F16>{}, std::integral_constant<int, YMI_F16>{});               \
        _try(std::integral_constant<int, YMI_F16>{}, std::integral_constant<int, YMI_F32>{});               \
        _try(std::integral_constant<int, YMI_BF16>{}, std::integral_constant<int, YMI_BF16>{});             \
        _try(std::integral_constant<int, YMI_BF16>{}, std::integral_constant<int, YMI_F32>{});              \
        if (!_done) { set_error("unsupported dtype pair %d -> %d", IDT_V, ODT_V); return YMI_EINVAL; }      \
    } while (0)

#include <type_traits>

extern "C" int ymi_nchw_to_nhwc(const void* x, int n, int c, int h, int w, int in_dtype, void* y, int y_cstride, int c_pad,
                                int out_dtype, void* stream) {
    YMI_REQUIRE(x && y && c_pad >= c && y_cstride >= c_pad, "ymi_nchw_to_nhwc: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)n * h * w;
    if (npix == 0) return YMI_OK;
    dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    YMI_DISPATCH_IO(nchw_to_nhwc_kernel, in_dtype, out_dtype, x, n, c, h, w, y, y_cstride, c_pad);
    return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int ymi_nhwc_to_nchw(const void* x, int x_cstride, int n, int c, int h, int w, int in_dtype, void* y, int out_dtype,
                                void* stream) {
    YMI_REQUIRE(x && y && x_cstride >= c, "ymi_nhwc_to_nchw: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)n * c * h * w;
    if (total == 0) return YMI_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    YMI_DISPATCH_IO(nhwc_to_nchw_kernel, in_dtype, out_dtype, x, x_cstride, n, c, h, w, y);
    return check_launch("nhwc_to_nchw_kernel");
}

extern "C" int ymi_spp_pool(void* buf, int n, int h, int w, int c, int cstride, int dtype, void* stream) {
    YMI_REQUIRE(buf && c % 8 == 0 && cstride >= 4 * c && cstride % 8 == 0, "ymi_spp_pool: c %% 8 == 0 and cstride >= 4c required");
    const int64_t total = (int64_t)n * h * w * (c / 8);
    if (total == 0) return YMI_OK;
    if (dtype == YMI_F32) {   // fp32 mode
        const size_t lds32 = (size_t)h * w * 16 * 2;
        if (lds32 <= 160 * 1024 - 512 && cstride % 4 == 0) {   // LDS cascade: one block per (image, 4 channels); else the direct kernel
            int nt = 256;
            while (nt < 1024 && (size_t)h * w > (size_t)2 * nt) nt *= 2;
            if (lds32 > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)spp_pool_f32_lds_kernel, (int)lds32); if (rc_lds != YMI_OK) return rc_lds; }
            hipLaunchKernelGGL(spp_pool_f32_lds_kernel, dim3((unsigned)(n * (c / 4))), dim3((unsigned)nt), lds32, (hipStream_t)stream, (float*)buf, h, w, c, cstride);
            return check_launch("spp_pool_f32_lds_kernel");
        }
        dim3 g32((unsigned)((2 * total + 255) / 256)), b32(256);
        hipLaunchKernelGGL(spp_pool_f32_kernel, g32, b32, 0, (hipStream_t)stream, (float*)buf, n, h, w, c, cstride);
        return check_launch("spp_pool_f32_kernel");
    }
    YMI_REQUIRE(dtype == YMI_F16 || dtype == YMI_BF16, "ymi_spp_pool: dtype must be F16/BF16/F32");
    // LDS cascade form: the plane of 8*G channels TWICE in LDS (the stage's plane, overwritten in place by its column pass, and the row-pass scratch;
    // three copies until round 3); G = 4 (64-byte runs per pixel) when it fits, else 2 (yolov5m/l at 1280x1280: 40x40 maps, 32-byte runs), else 1
    // (round 1 fell back to the 169-tap direct kernel there: 1.75 ms per step at C3; round 2: G = 1, 324 us)
    int G = (c % 32 == 0) ? 4 : ((c % 16 == 0) ? 2 : 1);
    if (G == 4 && (size_t)h * w * G * 16 * 2 > 160 * 1024 - 512) G = 2;
    if (G == 2 && (size_t)h * w * G * 16 * 2 > 160 * 1024 - 512) G = 1;
    if (const char* ge = getenv("YOLORT_AMD_SPP_G")) { const int g = atoi(ge); if (g == 1 || (g == 2 && G >= 2) || (g == 4 && G == 4)) G = g; }   // tuning aid
    const size_t lds = (size_t)h * w * G * 16 * 2;
    if (lds <= 160 * 1024 - 512) {
        // threads per block: enough for ~two pixels per thread and pass, 256 ... 1024 (YOLORT_AMD_SPP_NT: tuning aid)
        int nt = 256;
        while (nt < 1024 && (size_t)h * w * G > (size_t)2 * nt) nt *= 2;
        if (const char* te = getenv("YOLORT_AMD_SPP_NT")) { const int v = atoi(te); if (v == 256 || v == 512 || v == 1024) nt = v; }
        dim3 g((unsigned)(n * (c / (8 * G)))), b((unsigned)nt);
        auto launch = [&](auto kfn) -> int {
            if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)kfn, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
            hipLaunchKernelGGL(kfn, g, b, lds, (hipStream_t)stream, (uint16_t*)buf, h, w, c, cstride);
            return check_launch("spp_pool_lds_kernel");
        };
        if (dtype == YMI_F16) return G == 4 ? launch(spp_pool_lds_kernel<YMI_F16, 4>) : (G == 2 ? launch(spp_pool_lds_kernel<YMI_F16, 2>) : launch(spp_pool_lds_kernel<YMI_F16, 1>));
        return G == 4 ? launch(spp_pool_lds_kernel<YMI_BF16, 4>) : (G == 2 ? launch(spp_pool_lds_kernel<YMI_BF16, 2>) : launch(spp_pool_lds_kernel<YMI_BF16, 1>));
    }
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (dtype == YMI_F16) hipLaunchKernelGGL((spp_pool_kernel<YMI_F16>), grid, block, 0, (hipStream_t)stream, (uint16_t*)buf, n, h, w, c, cstride);
    else if (dtype == YMI_BF16) hipLaunchKernelGGL((spp_pool_kernel<YMI_BF16>), grid, block, 0, (hipStream_t)stream, (uint16_t*)buf, n, h, w, c, cstride);
    else { set_error("ymi_spp_pool: dtype must be F16/BF16"); return YMI_EINVAL; }
    return check_launch("spp_pool_kernel");
}

extern "C" int ymi_upsample2x(const void* x, int x_cstride, int n, int h, int w, int c, void* y, int y_cstride, int dtype, void* stream) {
    YMI_REQUIRE(x && y && c % 8 == 0 && x_cstride % 8 == 0 && y_cstride % 8 == 0, "ymi_upsample2x: channels/strides must be multiples of 8");
    YMI_REQUIRE(dtype == YMI_F16 || dtype == YMI_BF16 || dtype == YMI_F32, "ymi_upsample2x: dtype must be F16/BF16/F32");
    if (dtype == YMI_F32) { c *= 2; x_cstride *= 2; y_cstride *= 2; }   // fp32 parity mode: a typeless 16-byte copy, counted in 2-byte units
    const int64_t total = (int64_t)n * h * w * (c / 8);
    if (total == 0) return YMI_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL(upsample2x_kernel, grid, block, 0, (hipStream_t)stream, (const uint16_t*)x, x_cstride, n, h, w, c, (uint16_t*)y, y_cstride);
    return check_launch("upsample2x_kernel");
}

extern "C" int ymi_act(void* y, int y_cstride, int npix, int c, int dtype, int act, const void* res, int res_cstride, void* stream) {
    YMI_REQUIRE(y != nullptr && npix >= 0 && c > 0, "ymi_act: null buffer / bad extent");
    YMI_REQUIRE(act == YMI_ACT_HARDSWISH || act == YMI_ACT_LEAKY, "ymi_act: activation must be YMI_ACT_HARDSWISH or YMI_ACT_LEAKY (SiLU rides in the convolution's epilogue)");
    YMI_REQUIRE(dtype == YMI_F16 || dtype == YMI_BF16 || dtype == YMI_F32, "ymi_act: dtype must be F16/BF16/F32");
    const int lanes = dtype == YMI_F32 ? 4 : 8;   // elements per 16-byte packet
    YMI_REQUIRE(c % lanes == 0 && y_cstride % lanes == 0 && (res == nullptr || res_cstride % lanes == 0), "ymi_act: channels / strides must be multiples of %d", lanes);
    const int64_t total = (int64_t)npix * (c / lanes);
    if (total == 0) return YMI_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == YMI_F32) hipLaunchKernelGGL(act_f32_kernel, grid, block, 0, s, (float*)y, y_cstride, (int64_t)npix, c, act, (const float*)res, res_cstride);
    else if (dtype == YMI_F16) hipLaunchKernelGGL((act_kernel<YMI_F16>), grid, block, 0, s, (uint16_t*)y, y_cstride, (int64_t)npix, c, act, (const uint16_t*)res, res_cstride);
    else hipLaunchKernelGGL((act_kernel<YMI_BF16>), grid, block, 0, s, (uint16_t*)y, y_cstride, (int64_t)npix, c, act, (const uint16_t*)res, res_cstride);
    return check_launch("act_kernel");
}

extern "C" int ymi_copy_view(const void* x, int x_cstride, int npix, int c, void* y, int y_cstride, int dtype, void* stream) {
    YMI_REQUIRE(x && y && c % 8 == 0 && x_cstride % 8 == 0 && y_cstride % 8 == 0, "ymi_copy_view: channels/strides must be multiples of 8");
    YMI_REQUIRE(dtype == YMI_F16 || dtype == YMI_BF16 || dtype == YMI_F32, "ymi_copy_view: dtype must be F16/BF16/F32");
    if (dtype == YMI_F32) { c *= 2; x_cstride *= 2; y_cstride *= 2; }   // fp32 parity mode: typeless copy in 2-byte units
    const int64_t total = (int64_t)npix * (c / 8);
    if (total == 0) return YMI_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL(copy_view_kernel, grid, block, 0, (hipStream_t)stream, (const uint16_t*)x, x_cstride, (int64_t)npix, c, (uint16_t*)y, y_cstride);
    return check_launch("copy_view_kernel");
}
