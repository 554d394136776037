// TOOLS ONLY (never linked into the product): what the LDS and the matrix pipe of one MI355X CU deliver to the fragment loop of the resident-weights 3x3 kernels
// (round 4: tile 134's compute phase alone -- no patch loads, no stores -- takes 13.8 k cycles per tile for 2.3 k cycles of MFMA work per wave, and neither the
// prefetch depth of its LDS reads nor a second independent block changes that).  Each test runs 8 waves per CU (two per SIMD, as the kernels do) on every CU and
// reports cycles per wave-iteration from s_memtime:
//   0  ds_read_b128, lane-linear addresses (64 lanes x 16 B contiguous)          -> LDS bytes per clock per CU at its best
//   1  ds_read_b128 with tile 134's fragment addresses (128-byte slots, chunk swizzle, the nine taps)
//   2  v_mfma_f32_32x32x16_f16, two independent accumulator chains per wave, operands in registers -> MFMA issue rate
//   3  the unit of the kernel: two reads (pattern 1) + two MFMAs, reads PF = 3 units ahead, counted lgkmcnt
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/lds_mfma_bench tools/lds_mfma_bench.hip ; run: tools/_bin/lds_mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 200, UNITS = 36;

__device__ __forceinline__ int swz(int pr, int ci) { return (((pr >> 1) & 3) << 1) | ((ci >> 1) & 1); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void bench(unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int lane = threadIdx.x & 63, hi = lane >> 5, frow = lane & 31;
    for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = 0.001f * i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)sm;
    unsigned ea[9];
    for (int t = 0; t < 9; ++t) {
        if (MODE == 0) { ea[t] = base + lane * 16 + t * 1024; continue; }
        const int dy = t / 3, dx = t % 3, pr = 2 * (frow >> 3) + dy, col = 2 * (frow & 7) + dx, ci = col >> 1;
        ea[t] = base + (pr * 18 + ((col & 1) ? 10 : 0) + ci) * 128 + ((hi ^ swz(pr, ci)) * 16);
    }
    f16x8 w0 = {1, 2, 3, 4, 5, 6, 7, 8}, w1 = {2, 1, 2, 1, 2, 1, 2, 1};
    f32x16 acc0 = {}, acc1 = {};
    f16x8 fa[3][2] = {};
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (MODE <= 1) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                const unsigned a0 = ea[u / 4] ^ ((u % 4) << 5);
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[u % 3][0]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:18432" : "=v"(fa[u % 3][1]) : "v"(a0));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc0[0] += (float)fa[0][0][0] + (float)fa[1][1][1] + (float)fa[2][0][2];
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, w1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, w0, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned a0 = ea[u / 4] ^ ((u % 4) << 5);
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[u % 3][0]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:18432" : "=v"(fa[u % 3][1]) : "v"(a0));
            }
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                if (u + 2 < UNITS) {
                    const unsigned a0 = ea[(u + 2) / 4] ^ (((u + 2) % 4) << 5);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[(u + 2) % 3][0]) : "v"(a0));
                    asm volatile("ds_read_b128 %0, %1 offset:18432" : "=v"(fa[(u + 2) % 3][1]) : "v"(a0));
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                } else if (u + 1 < UNITS) {
                    asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, fa[u % 3][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, fa[u % 3][1], acc1, 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; out[512 + blockIdx.x] = r1 - r0; }
    if (acc0[0] + acc1[3] == 12345.678f) sink[0] = acc0[1];
}

template <int MODE>
static void run(const char* what, double bytes_per_iter_wave, double mfma_per_iter_wave) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 1024 * 8); hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench<MODE>, dim3(512), dim3(256), 40 * 1024, 0, d, sink);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0, rsum = 0; for (int i = 0; i < 512; ++i) { sum += (double)h[i]; rsum += (double)h[512 + i]; }
    const double cyc = sum / 512 / ITERS;       // s_memtime (shader clock) cycles per iteration of one wave, 8 waves per CU running
    const double ns = rsum / 512 / ITERS * 10;   // s_memrealtime: 100 MHz
    printf("%-66s %7.0f shader cycles (%6.0f ns, %4.0f MHz) per iter per wave;  per CU: %5.0f B/clk LDS, MFMA-pipe use %.2f\n", what, cyc, ns, cyc / ns * 1e3,
           8.0 * bytes_per_iter_wave / cyc, (2.0 * mfma_per_iter_wave * 32.0) / cyc);
    hipFree(d); hipFree(sink);
}

int main() {
    run<0>("ds_read_b128 lane-linear (72 reads / iter / wave)", 72.0 * 1024, 0);
    run<1>("ds_read_b128, tile 134's fragment addresses (72 reads)", 72.0 * 1024, 0);
    run<2>("v_mfma 32x32x16 f16, 2 chains / wave (72 MFMAs)", 0, 72);
    run<3>("36 units: 2 reads + 2 MFMAs, reads 2 units ahead, counted waits", 72.0 * 1024, 72);
    return 0;
}
