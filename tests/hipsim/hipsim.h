// hipsim -- a lane-accurate CPU runtime for the HIP kernels of yolort_amd/csrc (TEST INFRASTRUCTURE, never linked into the product).
//
// The kernel sources are compiled UNCHANGED as host C++ (clang++, x86) against this header instead of <hip/hip_runtime.h>:
// every lane of a block is a fiber (ucontext) scheduled round-robin on one thread, `threadIdx` & co. are globals the scheduler
// sets before it resumes a lane, LDS is an ordinary array, and the wave-collective instructions the kernels use are restated:
//   v_mfma_f32_32x32x16_{f16,bf16}   D[cout][pixel] += A[cout][k] B[k][pixel], fp32, k ascending; lane (hi, r) supplies row / pixel r,
//                                    k = 8*hi .. 8*hi+7 and receives acc[g*4 + e] = D[g*8 + hi*4 + e][r]   (cdna_hip_programming.md, MFMA layouts)
//   v_permlane32_swap                the upper half of the first operand is exchanged with the lower half of the second
//   v_readfirstlane, s_barrier, global_load_lds (lane l's 16 bytes land at the wave-uniform LDS base + 16*l)
// v_exp_f32 / v_rcp_f32 are exp2f / 1.0f/x here (the hardware's are ~1 ulp approximations): results are compared with torch
// within a tolerance, and -- the point of the exercise -- BETWEEN kernels bit for bit (fused vs separate launches), and every
// out-of-bounds LDS / global access, missed barrier (deadlock) or wrong lane mapping shows up on the CPU, without a GPU.
// Lanes of a wave do not run in lockstep: wave-private LDS exchanges are ordered by __builtin_amdgcn_wave_barrier() only.
// What it cannot show: timing, bank conflicts, register pressure, memory-model races between waves (lanes run to their next
// barrier in a fixed order), and the exact rounding of the MFMA's internal adder tree.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <cmath>
#include <functional>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

// ---- language keywords of HIP ----
#define __global__
#define __device__
#define __host__
#ifndef __shared__
#define __shared__   /* kernels with STATIC __shared__ arrays are built with -D__shared__=static from a copy whose `extern __shared__` lost the keyword (test_hipsim_kernels.py) */
#endif
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef struct hipsimStream* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct int2 { int x, y; };

namespace hipsim {

extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern int g_max_lds;              // largest dynamic-LDS size a launch asked for (checked by the driver against its arrays)
extern long g_launches;
extern const void* g_kernarg;       // the running launch's first kernel argument (kernels read a single by-value block through the kernarg segment pointer)

void yield();
void wave_sync();                  // every lane of the calling lane's wave
void block_sync();                 // __syncthreads
unsigned char* wave_deposit();     // 64 lanes x 64 bytes of exchange space of the calling lane's wave
void run_block(const std::function<void()>& body, dim3 grid, dim3 block, dim3 bidx);

inline int lane_id() { return (int)(g_threadIdx.x & 63); }

template <class T>
inline T readfirstlane(T v) {
    unsigned char* d = wave_deposit();
    if (lane_id() == 0) memcpy(d, &v, sizeof(T));
    wave_sync();
    T r;
    memcpy(&r, d, sizeof(T));
    wave_sync();
    return r;
}

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
inline u32x2_t permlane32_swap(uint32_t a, uint32_t b) {
    unsigned char* d = wave_deposit();
    const int l = lane_id();
    memcpy(d + l * 64, &a, 4);
    memcpy(d + l * 64 + 4, &b, 4);
    wave_sync();
    uint32_t ra = a, rb = b;
    if (l >= 32) memcpy(&ra, d + (l - 32) * 64 + 4, 4);   // upper half of a <- lower half of b
    else memcpy(&rb, d + (l + 32) * 64, 4);               // lower half of b <- upper half of a
    wave_sync();
    u32x2_t r = {ra, rb};
    return r;
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <class Frag>
inline f32x16_t mfma_32x32x16(Frag a, Frag b, f32x16_t c) {
    unsigned char* d = wave_deposit();
    const int l = lane_id(), hi = l >> 5, r = l & 31;
    float af[8], bf[8];
    for (int i = 0; i < 8; ++i) {
        af[i] = (float)a[i];
        bf[i] = (float)b[i];
    }
    memcpy(d + l * 64, af, 32);
    memcpy(d + l * 64 + 32, bf, 32);
    wave_sync();
    for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 4; ++e) {
            const int cout = g * 8 + hi * 4 + e;
            float acc = c[g * 4 + e];
            for (int k = 0; k < 16; ++k) {
                float av, bv;
                memcpy(&av, d + ((k >> 3) * 32 + cout) * 64 + (k & 7) * 4, 4);
                memcpy(&bv, d + ((k >> 3) * 32 + r) * 64 + 32 + (k & 7) * 4, 4);
                acc += av * bv;
            }
            c[g * 4 + e] = acc;
        }
    wave_sync();
    return c;
}

// every lane's value of v (all 64 lanes of the wave must take part: wave-uniform control flow around the call)
template <class T>
inline void wave_gather(T v, T* out64) {
    static_assert(sizeof(T) <= 64, "deposit slot");
    unsigned char* d = wave_deposit();
    memcpy(d + lane_id() * 64, &v, sizeof(T));
    wave_sync();
    for (int l = 0; l < 64; ++l) memcpy(&out64[l], d + l * 64, sizeof(T));
    wave_sync();
}
template <class T>
inline T shfl(T v, int src) {
    T all[64];
    wave_gather(v, all);
    return all[src & 63];
}
template <class T>
inline T shfl_up(T v, int d) {   // lanes below d keep their own value
    T all[64];
    wave_gather(v, all);
    const int l = lane_id();
    return l >= d ? all[l - d] : v;
}
inline unsigned long long ballot(bool p) {
    int all[64];
    wave_gather((int)p, all);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m |= (unsigned long long)(all[l] != 0) << l;
    return m;
}

// v_mfma_f32_32x32x2_f32: lane (hi, r) supplies A[row r][k = hi] and B[k = hi][column r]; per output an fmaf chain over k = 0, 1
inline f32x16_t mfma_32x32x2f32(float a, float b, f32x16_t c) {
    unsigned char* d = wave_deposit();
    const int l = lane_id(), hi = l >> 5, r = l & 31;
    memcpy(d + l * 64, &a, 4);
    memcpy(d + l * 64 + 4, &b, 4);
    wave_sync();
    for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 4; ++e) {
            const int row = g * 8 + hi * 4 + e;
            float acc = c[g * 4 + e];
            for (int k = 0; k < 2; ++k) {
                float av, bv;
                memcpy(&av, d + (k * 32 + row) * 64, 4);
                memcpy(&bv, d + (k * 32 + r) * 64 + 4, 4);
                acc = fmaf(av, bv, acc);
            }
            c[g * 4 + e] = acc;
        }
    wave_sync();
    return c;
}

inline void global_load_lds16(const void* g, void* lds_wave_uniform) { memcpy((unsigned char*)lds_wave_uniform + lane_id() * 16, g, 16); }

inline const void* first_arg_address() { return nullptr; }
template <class A0, class... A>
inline const void* first_arg_address(const A0& a0, const A&...) { return &a0; }
template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t lds, A... args) {
    if ((int)lds > g_max_lds) g_max_lds = (int)lds;
    ++g_launches;
    g_kernarg = first_arg_address(args...);
    static const bool trace = getenv("HIPSIM_TRACE") != nullptr;   // debugging aid: one line per launch
    if (trace) fprintf(stderr, "hipsim launch %ld: grid (%u, %u, %u) block %u lds %zu\n", g_launches, grid.x, grid.y, grid.z, block.x, lds);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) run_block([&]() { kernel(args...); }, grid, block, dim3(bx, by, bz));
}

}  // namespace hipsim

#define threadIdx hipsim::g_threadIdx
#define blockIdx hipsim::g_blockIdx
#define blockDim hipsim::g_blockDim
#define gridDim hipsim::g_gridDim

// ---- runtime API used by the launchers ----
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hipsim::launch(kernel, dim3(grid), dim3(block), (size_t)(lds), __VA_ARGS__)
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipsim"; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
template <class K>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) {
    const char* e = getenv("HIPSIM_BLOCKS_PER_CU");   // the launchers size their persistent grids as blocks-per-CU x 256 CUs
    *n = e ? atoi(e) : 1;
    return hipSuccess;
}
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s_, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s_, n); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
struct hipFuncAttributes { int numRegs; size_t sharedSizeBytes; int maxDynamicSharedSizeBytes; };
inline hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { memset(a, 0, sizeof(*a)); return hipSuccess; }

// ---- device builtins ----
#define __syncthreads() hipsim::block_sync()
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipsim::mfma_32x32x16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipsim::mfma_32x32x16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipsim::mfma_32x32x2f32(a, b, c)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipsim::permlane32_swap(a, b)
#define __builtin_amdgcn_readfirstlane(v) hipsim::readfirstlane(v)
#define __builtin_amdgcn_exp2f(v) exp2f(v)
#define __builtin_amdgcn_rcpf(v) (1.0f / (v))
// lanes of a wave do NOT run in lockstep here (each runs to its next collective): an exchange through LDS inside one wave needs a
// sync between its writes and its reads -- the kernels carry __builtin_amdgcn_wave_barrier() (a scheduling fence on the GPU) there
#define __builtin_amdgcn_wave_barrier() hipsim::wave_sync()
#define __builtin_amdgcn_s_barrier() hipsim::block_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipsim::global_load_lds16((const void*)(g), (void*)(l))
#define __shfl(v, src, ...) hipsim::shfl(v, (int)(src))
#define __shfl_xor(v, mask, ...) hipsim::shfl(v, hipsim::lane_id() ^ (int)(mask))
#define __ballot(p) hipsim::ballot((bool)(p))
#define __builtin_amdgcn_readlane(v, l) hipsim::shfl(v, (int)(l))
#define __shfl_up(v, d, ...) hipsim::shfl_up(v, (int)(d))
#define __ffsll(x) __builtin_ffsll(x)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __threadfence() {}   // one lane runs at a time and blocks run one after another: every earlier write is visible
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
#define __logf(x) logf(x)   /* fast-math intrinsics: the accurate libm forms (the head's pre-filter threshold carries a margin for exactly this) */
#define __expf(x) expf(x)
#define __fmul_rn(a, b) ((float)(a) * (float)(b))   /* the simulator build uses -ffp-contract=off: no fused multiply-add */
#define __fadd_rn(a, b) ((float)(a) + (float)(b))
#define __fsub_rn(a, b) ((float)(a) - (float)(b))
#define __fdiv_rn(a, b) ((float)(a) / (float)(b))
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }   // one thread runs at a time
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
#define __builtin_amdgcn_kernarg_segment_ptr() (hipsim::g_kernarg)
#define __umulhi(a, b) ((unsigned)(((uint64_t)(unsigned)(a) * (uint64_t)(unsigned)(b)) >> 32))
// HIP's device-side integer min / max overloads
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// inline assembly (waits, scheduling fences) has no meaning here: `asm volatile(...)` -> nothing.  (Function-like macro named
// after a keyword: legal only because every system header is included above.)
#define volatile(...)
#define asm
#define YMI_HIPSIM 1   // kernels that issue LDS reads through inline assembly (explicitly scheduled loops) take their plain C++ form here
