// Tuning aid: throughput of the global -> LDS DMA path (global_load_lds_dwordx4) on gfx950 as a function of
// access shape, residency (L2 / MALL / HBM), waves per CU and pieces in flight.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fill_bench.hip -o tools/_bin/fill_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const char* g, char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// mode 0: 1 KiB contiguous per piece; mode 1: 16 rows x 64 B (row stride `stride`), consecutive pieces walk along the row;
// mode 2: 8 rows x 128 B; mode 3: 4 rows x 256 B
template <int INFLIGHT>
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ src, size_t region_bytes, int pieces_per_wave, int mode, int stride, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    char* ring = smem + wave * (INFLIGHT + 1) * 1024;
    // this wave's region
    const size_t wid = (size_t)blockIdx.x * nwaves + wave;
    const size_t wave_bytes = (size_t)pieces_per_wave * 1024;
    const char* base = src + (wid * wave_bytes) % region_bytes;
    const bool rot = mode >= 4;   // modes 4..6 = modes 1..3 with a per-wave rotated walk along the row (no lockstep channel camping)
    if (rot) mode -= 3;
    const int rows = mode == 1 ? 16 : (mode == 2 ? 8 : (mode == 3 ? 4 : 1));
    const int row_bytes = 1024 / rows;
    const int lanes_per_row = row_bytes / 16;
    const int my_row = lane / lanes_per_row, my_chunk = lane % lanes_per_row;
    const int pieces_per_rowgroup = stride / row_bytes;   // pieces that walk along one group of rows
    int slot = 0;
    if (mode == 0) {
        const char* g = base + lane * 16;
        for (int p = 0; p < pieces_per_wave; ++p) {
            glds16(g, ring + slot * 1024);
            g += 1024;
            slot = slot == INFLIGHT ? 0 : slot + 1;
            wait_vmcnt<INFLIGHT>();
        }
    } else {
        const int ngroups = pieces_per_wave / pieces_per_rowgroup;
        const int ks0 = rot ? (int)(wid % pieces_per_rowgroup) : 0;
        const char* grow = base + (size_t)my_row * stride + my_chunk * 16;
        for (int rg = 0; rg < ngroups; ++rg) {
            int ks = ks0;
            for (int i = 0; i < pieces_per_rowgroup; ++i) {
                glds16(grow + ks * row_bytes, ring + slot * 1024);
                ks = ks + 1 == pieces_per_rowgroup ? 0 : ks + 1;
                slot = slot == INFLIGHT ? 0 : slot + 1;
                wait_vmcnt<INFLIGHT>();
            }
            grow += (size_t)rows * stride;
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (sink && threadIdx.x == 0 && smem[lane] == 123) sink[0] = 1;
}

int main(int argc, char** argv) {
    const size_t big = (size_t)1 << 30;
    char* buf; int* sink;
    hipMalloc(&buf, big + (1 << 20)); hipMemset(buf, 1, big); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-8s %-6s %-7s %-9s %-9s %10s %10s\n", "region", "mode", "waves", "blk/CU", "inflight", "TB/s", "B/clk/CU@2.1");
    const size_t regions[3] = {(size_t)2 << 20, (size_t)96 << 20, big};
    const int stride = argc > 1 ? atoi(argv[1]) : 512;
    const char* rn[3] = {"2MB", "96MB", "1GB"};
    for (int r = 0; r < 2; ++r)
        for (int mode = 0; mode < 7; ++mode)
            for (int cfg : {0, 1, 4}) {
                const int threads = cfg == 0 ? 256 : (cfg == 1 ? 256 : (cfg == 2 ? 512 : (cfg == 3 ? 1024 : 256)));
                const int bpc = cfg == 0 ? 1 : (cfg == 1 ? 2 : (cfg == 2 ? 1 : (cfg == 3 ? 1 : 4)));
                for (int inflight : {3, 12}) {
                    const int nwaves = threads / 64;
                    const int blocks = 256 * bpc;
                    const size_t total_target = (size_t)1 << 30;   // bytes moved per launch
                    int ppw = (int)(total_target / 1024 / ((size_t)blocks * nwaves));
                    ppw = ppw / 8 * 8;
                    const size_t lds = (size_t)nwaves * (inflight + 1) * 1024;
                    if (lds * bpc > 160 * 1024) continue;
                    auto launch = [&]() {
                        if (inflight == 3) hipLaunchKernelGGL(fill_kernel<3>, dim3(blocks), dim3(threads), lds, 0, buf, regions[r], ppw, mode, stride, sink);
                        else hipLaunchKernelGGL(fill_kernel<12>, dim3(blocks), dim3(threads), lds, 0, buf, regions[r], ppw, mode, stride, sink);
                    };
                    launch(); hipDeviceSynchronize();
                    hipEventRecord(e0, 0); launch(); launch(); hipEventRecord(e1, 0); hipDeviceSynchronize();
                    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
                    const double bytes = (double)ppw * 1024 * blocks * nwaves;
                    printf("%-8s %-6d %-7d %-9d %-9d %10.2f %10.1f\n", rn[r], mode, nwaves, bpc, inflight, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
                }
            }
    return 0;
}
