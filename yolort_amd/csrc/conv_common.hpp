// Shared pieces of the convolution kernels (conv_igemm.hip, conv3x3_halo.hip).
#pragma once
#include "common.hpp"

namespace ymi {

constexpr int BK = 32;          // k elements per main-loop step
constexpr int LDS_PITCH = 40;   // halfs per LDS row (32 + 8 pad) = 80 bytes
// v2 epilogue flavour: false = direct 16-byte stores from the MFMA layout after a permlane32 swap,
// true = stage the tile through LDS and write whole pixel rows (measured slower on yolov5s: -4%)
constexpr bool STAGED_EPILOGUE = false;

template <int DT>
struct Mfma;
template <>
struct Mfma<YMI_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mfma<YMI_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

struct ConvArgs {
    const uint16_t* x;
    const uint16_t* w;
    const float* bias;
    const int2* ktab;
    void* y;
    const uint16_t* res;
    int n, h, w_in, cin, x_cs;
    int ho, wo, cout, cout_pad, y_cs, res_cs;
    int sh, sw, ph, pw, k_pad;
    int act;
    int M;         // n*ho*wo
    int nblk_m, nblk_n;
    void* y2;      // second output view for couts >= split (0 = off)
    int y2_cs, split;
    const uint16_t* zeros;
    int x_zero_off;    // (zeros - x) in elements: out-of-range activation chunks read x + x_zero_off
    int kh, kw;
    unsigned magic_hw, magic_w;   // ceil(2^32 / (ho*wo)), ceil(2^32 / wo): exact floor-division for m < 2^31 / d ... see fast_div
    int debug;         // tuning aid: bit0 = skip LDS reads + MFMA, bit1 = skip operand loads (results are garbage)
};

// SiLU with hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each; the result is rounded to fp16/bf16)
// floor(n / d) for 0 <= n < 2^31 with magic = floor(2^32 / d) + 1: one mul_hi and a fix-up step
__device__ __forceinline__ int fast_div(int n, int d, unsigned magic) {
    int q = (int)__umulhi((unsigned)n, magic);
    if (q * d > n) --q;                 // magic over-estimates by at most one ...
    else if ((q + 1) * d <= n) ++q;     // ... and is clamped to 2^32-1 for d == 1 (under-estimates by one)
    return q;
}

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v)); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const uint16_t* g, uint16_t* lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}


// 3x3 stride-1 LDS-halo kernel (conv3x3_halo.hip); returns YMI_EINVAL when the shape does not apply
int conv3x3_halo_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// dedicated stem kernel (conv_stem.hip): 6x3 s(2,1) super-pixel form, input patch in LDS, weights in registers
int conv_stem_launch(const ConvArgs& a, int dtype, int out_dtype, hipStream_t s);

}  // namespace ymi
