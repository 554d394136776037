// Tuning aid: HBM streaming rate of a copy as a function of how a wave's 16-byte lanes map onto 128-byte lines (gfx950).
//   mode 0: lane l of a wave copies bytes [16 l, 16 l + 16) of a 1 KiB block (every instruction covers 8 full lines)
//   mode 1: the access shape of the convolution epilogue / streaming loads on 64-channel NHWC rows (128 B per pixel): a wave owns
//           32 pixels; instruction j (0..3) moves, for lane l, the 16 bytes at pixel (l & 31), chunk (l >> 5) + 2 j -- each
//           instruction touches 32 lines with 32 bytes each; the four together cover the lines
//   mode 2: loads as mode 0, stores as mode 1;   mode 3: loads as mode 1, stores as mode 0 (through registers; a wave's 4 KiB block
//           is the same bytes either way, only the order differs -- the copy is then a permutation inside the block)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/partial_line_bench.hip -o tools/_bin/partial_line_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const char* __restrict__ src, char* __restrict__ dst, size_t nblk4k) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nblk4k) return;
    const char* s = src + wave * 4096;
    char* d = dst + wave * 4096;
    u32x4 v[4];
    constexpr bool LP = MODE == 1 || MODE == 3, SP = MODE == 1 || MODE == 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int off = LP ? (lane & 31) * 128 + ((lane >> 5) + 2 * j) * 16 : j * 1024 + lane * 16;
        v[j] = *reinterpret_cast<const u32x4*>(s + off);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int off = SP ? (lane & 31) * 128 + ((lane >> 5) + 2 * j) * 16 : j * 1024 + lane * 16;
        *reinterpret_cast<u32x4*>(d + off) = v[j];
    }
}

int main() {
    const size_t bytes = (size_t)210 << 20;   // one 64-channel 160x160 batch-32 tensor
    const int NB = 8;                          // distinct buffer pairs: nothing stays in the MALL between launches
    char *src[NB], *dst[NB];
    for (int i = 0; i < NB; ++i) { hipMalloc(&src[i], bytes); hipMalloc(&dst[i], bytes); hipMemset(src[i], i + 1, bytes); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t nblk = bytes / 4096;
    const dim3 grid((unsigned)((nblk + 3) / 4)), block(256);
    const char* names[4] = {"loads full-line, stores full-line", "loads 32 B x 32 lines, stores 32 B x 32 lines", "loads full-line, stores 32 B x 32 lines", "loads 32 B x 32 lines, stores full-line"};
    for (int mode = 0; mode < 4; ++mode) {
        auto launch = [&](int i) {
            if (mode == 0) hipLaunchKernelGGL(copy_kernel<0>, grid, block, 0, 0, src[i], dst[i], nblk);
            else if (mode == 1) hipLaunchKernelGGL(copy_kernel<1>, grid, block, 0, 0, src[i], dst[i], nblk);
            else if (mode == 2) hipLaunchKernelGGL(copy_kernel<2>, grid, block, 0, 0, src[i], dst[i], nblk);
            else hipLaunchKernelGGL(copy_kernel<3>, grid, block, 0, 0, src[i], dst[i], nblk);
        };
        for (int i = 0; i < NB; ++i) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 3; ++r) for (int i = 0; i < NB; ++i) launch(i);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3 * NB;
        printf("mode %d (%s): %.1f us per launch, %.0f MB read + %.0f MB written -> %.2f TB/s\n", mode, names[mode], ms * 1e3, bytes / 1e6, bytes / 1e6, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}
