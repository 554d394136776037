// Shared helpers for the gfx950 kernels of libyolort_amd.so (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/yolort_amd.h"

namespace ymi {

void set_error(const char* fmt, ...);

#define YMI_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ymi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return YMI_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define YMI_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            ymi::set_error(__VA_ARGS__);    \
            return YMI_EINVAL;              \
        }                                   \
    } while (0)

// Opt-in to > 64 KiB of dynamic LDS: a per-function (and per-device) attribute.  It is set ONCE per (kernel, device) and
// remembered -- calling hipFuncSetAttribute on every launch sat on the enqueue path of every big-tile conv.
int allow_big_lds(const void* kernel_fn, int bytes);
// Tuning knob (YOLORT_AMD_LDS_FLOOR_KB, default 0): minimum dynamic LDS a conv launch requests.  A floor of 81 KiB caps a
// kernel at ONE block per CU, so that blocks of OTHER batches' kernels (other streams, other phases) share the CU instead of
// a second, phase-locked block of the same kernel.
// (measured r02i, C2 bs 32, 4-6 batches in flight: 19.3-19.5 k img/s at 0, 18.6-19.0 k at 54 KiB, 15.7-15.9 k at 81 KiB: the default stays 0)
size_t lds_floor_bytes();
// Fused detection head: split the cout axis by anchor (three blocks per 128-pixel tile, >= 3 waves per SIMD) instead of one
// 400-register wave per 32 pixels x 3 anchors.  YOLORT_AMD_HEAD_SPLIT=0 selects the unsplit form (A/B runs).
bool head_anchor_split();

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return YMI_EHIP;
    }
    return YMI_OK;
}

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- scalar conversions (round-to-nearest-even) ----
__device__ __forceinline__ float h2f(uint16_t v) {
    f16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
__device__ __forceinline__ uint16_t f2h(float f) {
    f16 h = (f16)f;
    uint16_t v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <int DT>
__device__ __forceinline__ float load_elem(const void* p, int64_t i) {
    if constexpr (DT == YMI_F16) return h2f(((const uint16_t*)p)[i]);
    else if constexpr (DT == YMI_BF16) return bf2f(((const uint16_t*)p)[i]);
    else if constexpr (DT == YMI_F32) return ((const float*)p)[i];
    else return (float)((const uint8_t*)p)[i] / 255.0f;   // YMI_U8 / YMI_U8_HWC: true division, like the reference's `read_image(...) / 255.0`
}
// two fp32 values -> one dword of two 16-bit values with the hardware pair conversion (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: round-to-nearest-even like f2h / f2bf;
// the software f2bf is 6-7 VALU instructions per value)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ uint32_t cvt_pk16(f32x2 v) {
    uint32_t u;
    if constexpr (DT == YMI_F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = __builtin_convertvector(v, h2);
        __builtin_memcpy(&u, &h, 4);
    } else {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const b2 b = __builtin_convertvector(v, b2);
        __builtin_memcpy(&u, &b, 4);
    }
    return u;
}
template <int DT>
__device__ __forceinline__ uint16_t to16(float v) {
    if constexpr (DT == YMI_F16) return f2h(v);
    else return f2bf(v);
}
template <int DT>
__device__ __forceinline__ float from16(uint16_t v) {
    if constexpr (DT == YMI_F16) return h2f(v);
    else return bf2f(v);
}

// element index of channel c at (y, x) of an h x w image: planar (3, h, w), or interleaved (h, w, 3) for YMI_U8_HWC
template <int DT>
__device__ __forceinline__ int64_t src_index(int c, int y, int x, int w, int64_t plane) {
    if constexpr (DT == YMI_U8_HWC) return ((int64_t)y * w + x) * 3 + c;
    else return c * plane + (int64_t)y * w + x;
}

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// XCD-aware, bijective block remap (cdna_hip_programming.md T1): consecutive logical ids share one
// XCD's L2 (block b is dispatched to XCD b % 8).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace ymi
