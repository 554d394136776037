// HBM-bound helper kernels of the conv stack edges: letterbox, layout changes, SPP max-pool
// pyramid, nearest x2 upsample and channel-slice copy.  All move 16 bytes per lane where the
// layout allows (cdna_hip_programming.md G13) and are launched with >> 256 workgroups.
#include "common.hpp"

namespace ymi {

// ---------------------------------------------------------------------------------------------
// Letterbox.  Replaces yolort/models/transform.py:53-97 (bilinear resize; ATen
// upsample_bilinear2d semantics: align_corners=False, scale recomputed as in/out, src clamped at
// 0, x1 = min(x0+1, in-1), fp32 lerp) and transform.py:297-330 (fill + centred copy).
// One thread per output pixel; up to LB_MAX images per launch (descriptors travel as kernargs so
// there is no device-side pointer table to allocate).
// ---------------------------------------------------------------------------------------------
constexpr int LB_MAX = 32;
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
            const int64_t b = c * plane;
            const float p00 = load_elem<IDT>(a.img[img], b + (int64_t)y0 * win + x0);
            const float p01 = load_elem<IDT>(a.img[img], b + (int64_t)y0 * win + x1);
            const float p10 = load_elem<IDT>(a.img[img], b + (int64_t)y1 * win + x0);
            const float p11 = load_elem<IDT>(a.img[img], b + (int64_t)y1 * win + x1);
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
template <int IDT, int ODT>
__global__ __launch_bounds__(256) void letterbox_copy_kernel(const LetterboxArgs a) {
    const int wq = a.wb / 4;
    const int64_t total = (int64_t)a.n * a.hb * wq;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int xq = (int)(gid % wq);
    const int64_t t = gid / wq;
    const int y = (int)(t % a.hb);
    const int img = (int)(t / a.hb);
    const int hin = a.geom[img][0], win = a.geom[img][1];
    const int yy = y - a.geom[img][4];
    const int x0 = xq * 4 - a.geom[img][5];
    float v[4][3];
    const bool row_in = (unsigned)yy < (unsigned)hin;
    const bool all_in = row_in && x0 >= 0 && x0 + 3 < win && (win % 4 == 0) && (a.geom[img][5] % 4 == 0);
    if (all_in && (IDT == YMI_F16 || IDT == YMI_BF16)) {
        const int64_t plane = (int64_t)hin * win;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const u32x2 p = *reinterpret_cast<const u32x2*>((const uint16_t*)a.img[img] + c * plane + (int64_t)yy * win + x0);
            v[0][c] = from16<IDT == YMI_BF16 ? YMI_BF16 : YMI_F16>((uint16_t)(p[0] & 0xffff));
            v[1][c] = from16<IDT == YMI_BF16 ? YMI_BF16 : YMI_F16>((uint16_t)(p[0] >> 16));
            v[2][c] = from16<IDT == YMI_BF16 ? YMI_BF16 : YMI_F16>((uint16_t)(p[1] & 0xffff));
            v[3][c] = from16<IDT == YMI_BF16 ? YMI_BF16 : YMI_F16>((uint16_t)(p[1] >> 16));
        }
    } else {
        const int64_t plane = (int64_t)hin * win;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xx = x0 + i;
            const bool in = row_in && (unsigned)xx < (unsigned)win;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[i][c] = in ? load_elem<IDT>(a.img[img], c * plane + (int64_t)yy * win + xx) : a.fill;
        }
    }
    if constexpr (ODT == YMI_F32) {
        float* o = (float*)a.out + (((int64_t)img * a.hb + y) * a.wb + xq * 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[4 * i] = v[i][0]; o[4 * i + 1] = v[i][1]; o[4 * i + 2] = v[i][2]; o[4 * i + 3] = 0.f; }
    } else {
        uint16_t* o = (uint16_t*)a.out + (((int64_t)img * a.hb + y) * a.wb + xq * 4) * 4;
        u32x4 lo, hi;
        lo[0] = (uint32_t)to16<ODT>(v[0][0]) | ((uint32_t)to16<ODT>(v[0][1]) << 16);
        lo[1] = (uint32_t)to16<ODT>(v[0][2]);
        lo[2] = (uint32_t)to16<ODT>(v[1][0]) | ((uint32_t)to16<ODT>(v[1][1]) << 16);
        lo[3] = (uint32_t)to16<ODT>(v[1][2]);
        hi[0] = (uint32_t)to16<ODT>(v[2][0]) | ((uint32_t)to16<ODT>(v[2][1]) << 16);
        hi[1] = (uint32_t)to16<ODT>(v[2][2]);
        hi[2] = (uint32_t)to16<ODT>(v[3][0]) | ((uint32_t)to16<ODT>(v[3][1]) << 16);
        hi[3] = (uint32_t)to16<ODT>(v[3][2]);
        *reinterpret_cast<u32x4*>(o) = lo;
        *reinterpret_cast<u32x4*>(o + 8) = hi;
    }
}

template <int IDT>
static int letterbox_dispatch(const LetterboxArgs& a, int out_dtype, hipStream_t s) {
    bool identity = a.c_out == 4 && a.wb % 4 == 0;
    for (int i = 0; i < a.n && identity; ++i) identity = a.geom[i][0] == a.geom[i][2] && a.geom[i][1] == a.geom[i][3];
    if (identity) {   // no resampling anywhere in this launch: interleave-copy kernel (bit-identical results)
        const int64_t tq = (int64_t)a.n * a.hb * (a.wb / 4);
        dim3 gq((unsigned)((tq + 255) / 256)), bq(256);
        switch (out_dtype) {
            case YMI_F16: hipLaunchKernelGGL((letterbox_copy_kernel<IDT, YMI_F16>), gq, bq, 0, s, a); break;
            case YMI_BF16: hipLaunchKernelGGL((letterbox_copy_kernel<IDT, YMI_BF16>), gq, bq, 0, s, a); break;
            case YMI_F32: hipLaunchKernelGGL((letterbox_copy_kernel<IDT, YMI_F32>), gq, bq, 0, s, a); break;
            default: set_error("ymi_letterbox: bad out_dtype %d", out_dtype); return YMI_EINVAL;
        }
        return check_launch("letterbox_copy_kernel");
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

// LDS cascade form of the same pyramid: one block per (image, 8-channel chunk).  The h x w plane of
// those 8 channels is staged in LDS as fp32 and three 5x5 max stages are applied back to back, each
// separable (row pass then column pass): mp9 = mp5(mp5(x)), mp13 = mp5(mp9) -- the SPPF identity of
// the reference (common.py:196), exact because max is exact.  10 taps per stage instead of a
// 169-tap window, and the plane is read from HBM once.
template <int DT>
__global__ __launch_bounds__(256) void spp_pool_lds_kernel(uint16_t* buf, int h, int w, int c, int cs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw = h * w;
    float* A = sm;
    float* B = sm + (size_t)hw * 8;
    float* Cb = sm + (size_t)hw * 16;
    const int c8 = c / 8;
    const int img = blockIdx.x / c8, cc = (blockIdx.x % c8) * 8;
    uint16_t* base = buf + (int64_t)img * hw * cs + cc;
    for (int p = threadIdx.x; p < hw; p += 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + (int64_t)p * cs);
#pragma unroll
        for (int e = 0; e < 8; ++e) A[p * 8 + e] = from16<DT>((uint16_t)((v[e >> 1] >> ((e & 1) * 16)) & 0xffff));
    }
    __syncthreads();
    float* src = A;
    float* dst = Cb;
    for (int stage = 0; stage < 3; ++stage) {
        for (int p = threadIdx.x; p < hw; p += 256) {  // row pass: src -> B
            const int y = p / w, x = p - y * w;
            const int x0 = x - 2 < 0 ? 0 : x - 2, x1 = x + 2 > w - 1 ? w - 1 : x + 2;
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
            for (int xx = x0; xx <= x1; ++xx)
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], src[(y * w + xx) * 8 + e]);
#pragma unroll
            for (int e = 0; e < 8; ++e) B[p * 8 + e] = m[e];
        }
        __syncthreads();
        for (int p = threadIdx.x; p < hw; p += 256) {  // column pass: B -> dst, and out to HBM
            const int y = p / w, x = p - y * w;
            const int y0 = y - 2 < 0 ? 0 : y - 2, y1 = y + 2 > h - 1 ? h - 1 : y + 2;
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
            for (int yy = y0; yy <= y1; ++yy)
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], B[(yy * w + x) * 8 + e]);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (uint32_t)to16<DT>(m[2 * e]) | ((uint32_t)to16<DT>(m[2 * e + 1]) << 16);
            *reinterpret_cast<u32x4*>(base + (int64_t)p * cs + (stage + 1) * c) = o;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[p * 8 + e] = m[e];
        }
        __syncthreads();
        float* t = src;  // ping-pong A <-> Cb (B is the row-pass scratch)
        src = dst;
        dst = t;
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
    YMI_REQUIRE(dtype == YMI_F16 || dtype == YMI_BF16, "ymi_spp_pool: dtype must be F16/BF16");
    const size_t lds = (size_t)h * w * 8 * 4 * 3;
    if (lds <= 160 * 1024 - 512) {  // whole plane of 8 channels fits the 160 KB LDS three times
        dim3 g((unsigned)(n * (c / 8))), b(256);
        if (dtype == YMI_F16) {
            if (lds > 64 * 1024) YMI_CHECK_HIP(hipFuncSetAttribute((const void*)spp_pool_lds_kernel<YMI_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((spp_pool_lds_kernel<YMI_F16>), g, b, lds, (hipStream_t)stream, (uint16_t*)buf, h, w, c, cstride);
        } else {
            if (lds > 64 * 1024) YMI_CHECK_HIP(hipFuncSetAttribute((const void*)spp_pool_lds_kernel<YMI_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((spp_pool_lds_kernel<YMI_BF16>), g, b, lds, (hipStream_t)stream, (uint16_t*)buf, h, w, c, cstride);
        }
        return check_launch("spp_pool_lds_kernel");
    }
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (dtype == YMI_F16) hipLaunchKernelGGL((spp_pool_kernel<YMI_F16>), grid, block, 0, (hipStream_t)stream, (uint16_t*)buf, n, h, w, c, cstride);
    else if (dtype == YMI_BF16) hipLaunchKernelGGL((spp_pool_kernel<YMI_BF16>), grid, block, 0, (hipStream_t)stream, (uint16_t*)buf, n, h, w, c, cstride);
    else { set_error("ymi_spp_pool: dtype must be F16/BF16"); return YMI_EINVAL; }
    return check_launch("spp_pool_kernel");
}

extern "C" int ymi_upsample2x(const void* x, int x_cstride, int n, int h, int w, int c, void* y, int y_cstride, int dtype, void* stream) {
    YMI_REQUIRE(x && y && c % 8 == 0 && x_cstride % 8 == 0 && y_cstride % 8 == 0, "ymi_upsample2x: channels/strides must be multiples of 8");
    YMI_REQUIRE(dtype == YMI_F16 || dtype == YMI_BF16, "ymi_upsample2x: dtype must be F16/BF16");
    const int64_t total = (int64_t)n * h * w * (c / 8);
    if (total == 0) return YMI_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL(upsample2x_kernel, grid, block, 0, (hipStream_t)stream, (const uint16_t*)x, x_cstride, n, h, w, c, (uint16_t*)y, y_cstride);
    return check_launch("upsample2x_kernel");
}

extern "C" int ymi_copy_view(const void* x, int x_cstride, int npix, int c, void* y, int y_cstride, int dtype, void* stream) {
    YMI_REQUIRE(x && y && c % 8 == 0 && x_cstride % 8 == 0 && y_cstride % 8 == 0, "ymi_copy_view: channels/strides must be multiples of 8");
    YMI_REQUIRE(dtype == YMI_F16 || dtype == YMI_BF16, "ymi_copy_view: dtype must be F16/BF16");
    const int64_t total = (int64_t)npix * (c / 8);
    if (total == 0) return YMI_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL(copy_view_kernel, grid, block, 0, (hipStream_t)stream, (const uint16_t*)x, x_cstride, (int64_t)npix, c, (uint16_t*)y, y_cstride);
    return check_launch("copy_view_kernel");
}
