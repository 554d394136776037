// fp32 PARITY MODE convolution (gfx950): fp32 NHWC activations, fp32 folded weights, exact fp32 arithmetic on
// v_mfma_f32_32x32x2_f32 (an fmaf chain per output, MI355X_MICROARCH.md "Matrix cores": f32-input MFMA runs at the
// fp32 vector rate, 157 TF).  This is the mode in which the whole HIP path reproduces the fp32 CPU reference to
// rounding-order accuracy (north-star tolerance: boxes within 1e-3 IoU, equal labels); the production path stores
// fp16/bf16 and runs the 16-bit MFMA kernels of conv_igemm.hip.
//
// Same GEMM view and accumulator layout as the 16-bit kernels (D[cout][pixel], lane l owns pixel l & 31, register
// g*4+e holds cout row g*8 + 4*(l>>5) + e), same descriptor (ymi_conv_desc with dtype = out_dtype = YMI_F32), same
// im2col table.  Block = 4 waves, tile 128 pixels x 64 couts, BK = 8 (one table chunk); operands are staged
// global -> registers -> LDS (k-major, conflict-free both ways), double buffered with one barrier per step.
//
// Replaces yolort/v5/models/common.py:69-70 (Conv.forward), :115-116 (Bottleneck residual) and
// yolort/models/box_head.py:36,74 (head conv) -- in fp32, like the reference's CPU path.
#include "conv_common.hpp"

namespace ymi {

constexpr int F_BM = 128, F_BN = 64, F_BK = 8;
constexpr int F_LDA = F_BM + 4, F_LDW = F_BN + 4;   // k-major rows; +4 keeps the two half-rows of a store on distinct banks

__device__ __forceinline__ float silu_exact(float v) { return v / (1.0f + expf(-v)); }   // torch CPU: x / (1 + exp(-x))

template <bool IS1X1>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvArgs a) {
    __shared__ float As[2][F_BK][F_LDA];
    __shared__ float Ws[2][F_BK][F_LDW];
    const float* __restrict__ X = reinterpret_cast<const float*>(a.x);
    const float* __restrict__ Wt = reinterpret_cast<const float*>(a.w);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = (wave >> 1) * 64, wave_n = (wave & 1) * 32;
    const int lb = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
    const int bm = lb / a.nblk_n, bn = lb % a.nblk_n;
    const int m0 = bm * F_BM, n0 = bn * F_BN;

    // gather geometry: thread loads floats [half*4, half*4+4) of chunk `step` for pixel row (tid >> 1)
    const int row = tid >> 1, half = tid & 1;
    const int m = m0 + row;
    const bool m_in = m < a.M;
    int64_t a_base = 0;
    int iy0 = 0, ix0 = 0;
    if (m_in) {
        const int hw = a.ho * a.wo;
        const int img = m / hw, rem = m - img * hw;
        const int oy = rem / a.wo, ox = rem - oy * a.wo;
        iy0 = oy * a.sh - a.ph;
        ix0 = ox * a.sw - a.pw;
        a_base = (((int64_t)img * a.h + iy0) * a.w_in + ix0) * a.x_cs;
    }
    const int wrow = n0 + row;                       // threads 0..127 load the weight tile (64 rows x 8)
    const bool w_in = tid < 128 && wrow < a.cout_pad;
    const int nsteps = a.k_pad / F_BK;

    f32x4 ra, rw;
    auto load_regs = [&](int step) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ra = z;
        rw = z;
        int koff, dy = 0, dx = 0;
        bool ok = m_in;
        if constexpr (IS1X1) {
            koff = step * F_BK;
            ok = ok && koff < a.cin;
        } else {
            const int2 t = a.ktab[step];
            koff = t.x;
            ok = ok && t.y >= 0;
            dy = t.y >> 16;
            dx = t.y & 0xffff;
            ok = ok && ((unsigned)(iy0 + dy) < (unsigned)a.h) && ((unsigned)(ix0 + dx) < (unsigned)a.w_in);
        }
        if (ok) ra = *reinterpret_cast<const f32x4*>(X + a_base + koff + half * 4);
        if (w_in) rw = *reinterpret_cast<const f32x4*>(Wt + (int64_t)wrow * a.k_pad + step * F_BK + half * 4);
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) As[buf][half * 4 + e][row] = ra[e];
        if (tid < 128) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[buf][half * 4 + e][row] = rw[e];
        }
    };

    // Two-level (blocked) summation: a chain of K sequential fp32 FMAs has a rounding error ~ sqrt(K) ulp, several times
    // what the CPU reference's vectorised / blocked sums produce, and the synthetic test network amplifies every ulp ~2500x
    // on its way to the logits.  Partial sums over 64 k-elements go to `part`, which is folded into `acc` every 8 steps:
    // error ~ sqrt(64) + sqrt(K / 64) ulp.
    f32x16 acc[2], part[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; part[j][r] = 0.f; }

    const int fcol = lane & 31, hi = lane >> 5;
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < nsteps) load_regs(step + 1);
#pragma unroll
        for (int s = 0; s < F_BK / 2; ++s) {
            const int k = 2 * s + hi;
            const float wf = Ws[cur][k][wave_n + fcol];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float xf = As[cur][k][wave_m + j * 32 + fcol];
                part[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, xf, part[j], 0, 0, 0);
            }
        }
        if ((step & 7) == 7 || step + 1 == nsteps) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[j][r] += part[j][r]; part[j][r] = 0.f; }
        }
        if (step + 1 < nsteps) store_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, exact SiLU, residual (added after the activation), fp32 stores ----
    const float* __restrict__ R = reinterpret_cast<const float*>(a.res);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t mo = (int64_t)m0 + wave_m + j * 32 + fcol;
        if (mo >= a.M) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = n0 + wave_n + g * 8 + hi * 4;
            if (co >= a.cout) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[j][g * 4 + e] + (co + e < a.cout_pad ? a.bias[co + e] : 0.f);
                if (a.act == YMI_ACT_SILU) t = silu_exact(t);
                if (R != nullptr && co + e < a.cout) t += R[mo * a.res_cs + co + e];
                v[e] = t;
            }
            float* yp;
            int cs;
            if (a.split > 0 && co >= a.split) { yp = reinterpret_cast<float*>(a.y2) + mo * a.y2_cs + (co - a.split); cs = a.y2_cs; }
            else { yp = reinterpret_cast<float*>(a.y) + mo * a.y_cs + co; cs = a.y_cs; }
            if (co + 3 < a.cout && (cs & 3) == 0 && ((a.split & 3) == 0)) {
                f32x4 o = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(yp) = o;
            } else {
                for (int e = 0; e < 4 && co + e < a.cout; ++e) yp[e] = v[e];
            }
        }
    }
}

int conv_f32_launch(const ConvArgs& a0, bool is1x1, hipStream_t s) {
    ConvArgs a = a0;
    YMI_REQUIRE(a.chain_w == nullptr && a.up2 == 0, "ymi_conv2d: the fp32 parity kernel has no chained conv / upsampled second output");
    YMI_REQUIRE(a.x_cs % 4 == 0 && a.k_pad % F_BK == 0, "ymi_conv2d (fp32): x_cstride %% 4 and k_pad %% 8 must be 0");
    YMI_REQUIRE(a.split == 0 || a.split % 4 == 0, "ymi_conv2d (fp32): cout_split must be a multiple of 4");
    a.nblk_m = cdiv(a.M, F_BM);
    a.nblk_n = cdiv(a.cout_pad, F_BN);
    dim3 grid(a.nblk_m * a.nblk_n), block(256);
    if (is1x1) hipLaunchKernelGGL((conv_f32_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_f32_kernel<false>), grid, block, 0, s, a);
    return check_launch("conv_f32_kernel");
}

}  // namespace ymi
