// Shared pieces of the convolution kernels (conv_igemm.hip, conv3x3_halo.hip).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace ymi {

constexpr int BK = 32;          // k elements per main-loop step
constexpr int LDS_PITCH = 40;   // halfs per LDS row (32 + 8 pad) = 80 bytes

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
    const uint16_t* chain_w;   // chained 1x1 conv on the first chain_k output channels (see ymi_conv_desc.chain_w); NULL = off
    const float* chain_bias;
    void* chain_y;
    int chain_k, chain_cout, chain_y_cs;
    const uint16_t* chain_x2;  // second K range of the chained conv: chain_k2 channels per pixel read from this view (NULL / 0 = none)
    int chain_x2_cs, chain_k2;
    int up2;       // 1: y2 is an (n, 2ho, 2wo) view receiving every output channel nearest-upsampled x2 (split == 0)
    const uint16_t* zeros;
    int x_zero_off;    // (zeros - x) in elements: out-of-range activation chunks read x + x_zero_off
    int kh, kw;
    unsigned magic_hw, magic_w;   // ceil(2^32 / (ho*wo)), ceil(2^32 / wo): exact floor-division for m < 2^31 / d ... see fast_div
    int debug;         // tuning aid: bit0 = skip LDS reads + MFMA, bit1 = skip operand loads, bit2 = head: no decode, bit4 = first K step only (results are garbage)
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

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N-1
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// tuning aid (-DYMI_SETPRIO=n, never in the shipped build): raise the wave's issue priority around its MFMA groups so that the partner wave's DMA issue /
// fragment reads fill the gaps instead of delaying the matrix pipe (measured: profiles/r04t_setprio.txt)
#ifdef YMI_SETPRIO
#define YMI_PRIO_HI() __builtin_amdgcn_s_setprio(YMI_SETPRIO)
#define YMI_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define YMI_PRIO_HI() ((void)0)
#define YMI_PRIO_LO() ((void)0)
#endif
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const uint16_t* g, uint16_t* lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}


// ---------------------------------------------------------------------------------------------------
// Shared epilogue of the convolution kernels.  Every kernel computes D[cout][pixel] 32x32 sub-tiles with
// the swapped MFMA: lane l owns pixel column (l & 31); accumulator register g*4+e holds cout row
// g*8 + 4*(l>>5) + e of the sub-tile.
//   * the bias is folded into the accumulator INIT (loaded once at kernel entry, its latency hidden behind
//     the geometry math and the DMA prologue) -- a bias load per sub-tile group in the epilogue costs one
//     serialized L2 round trip each (16 per wave for a 64x64 wave tile: measured ~2x the epilogue's math)
//   * the residual (Bottleneck shortcut) of a whole wave tile is fetched in one batch before the math
// ---------------------------------------------------------------------------------------------------
// The bias loads are issued BEHIND the prologue's LDS-DMA and consumed (init_acc) right before the main loop: issued first,
// their first use drained the vector-memory counter before any operand DMA had been sent -- every block began with two cold
// memory latencies in series (bias, then operands; 2.9 k cycles from block entry to the first loop iteration).  (A scalar
// s_load_dwordx8 form was tried: 16 x TN SGPRs at once spill the scalar file of the TN >= 2 kernels.)
template <int TN>
__device__ __forceinline__ void load_bias(const ConvArgs& a, int cbase0, int hi, f32x4 (&b)[TN][4]) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = cbase0 + i * 32 + g * 8 + hi * 4;
            b[i][g] = *reinterpret_cast<const f32x4*>(a.bias + (co < a.cout_pad ? co : 0));   // rows past cout_pad are never stored
        }
}

template <int TN, int TM>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[TN][TM], const f32x4 (&b)[TN][4]) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][g * 4 + e] = b[i][g][e];
}

__device__ __forceinline__ void load_residual(const ConvArgs& a, int64_t m, bool m_ok, int cbase, int hi, u32x2 (&r)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co = cbase + g * 8 + hi * 4;
        u32x2 z = {0u, 0u};
        r[g] = z;
        if (m_ok && co < a.cout) r[g] = *reinterpret_cast<const u32x2*>(a.res + m * a.res_cs + co);
    }
}

// activation (+ residual) + conversion + store of one 32x32 sub-tile; cbase (first cout of the sub-tile) is wave-uniform
template <int DT, int ODT, bool RES>
__device__ __forceinline__ void finish_subtile(const ConvArgs& a, const f32x16& acc, int64_t m, bool m_ok, int cbase, int hi, const u32x2 (&r)[4],
                                               int64_t m_up = 0, u32x4* frag_out = nullptr) {
    float v[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = acc[g * 4 + e];
            if (a.act == YMI_ACT_SILU) t = silu(t);
            v[g][e] = t;
        }
        if constexpr (RES) {
            v[g][0] += from16<DT>((uint16_t)(r[g][0] & 0xffff));
            v[g][1] += from16<DT>((uint16_t)(r[g][0] >> 16));
            v[g][2] += from16<DT>((uint16_t)(r[g][1] & 0xffff));
            v[g][3] += from16<DT>((uint16_t)(r[g][1] >> 16));
        }
    }
    if constexpr (ODT == YMI_F32) {
        if (!m_ok) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = cbase + g * 8 + hi * 4;
            if (co >= a.cout) continue;
            float* yp = reinterpret_cast<float*>(a.y) + m * a.y_cs + co;
            if (co + 3 < a.cout) {
                f32x4 o = {v[g][0], v[g][1], v[g][2], v[g][3]};
                *reinterpret_cast<f32x4*>(yp) = o;
            } else {
                for (int e = 0; e < 4 && co + e < a.cout; ++e) yp[e] = v[g][e];
            }
        }
    } else {
        uint32_t pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pk[g][0] = cvt_pk16<DT>(f32x2{v[g][0], v[g][1]});   // hardware pair conversion (round 4; the lean path's: same rounding as to16 -- bf16 in software is ~7 VALU per value)
            pk[g][1] = cvt_pk16<DT>(f32x2{v[g][2], v[g][3]});
        }
        const bool wide = (cbase + 32 <= a.cout) && ((a.split & 7) == 0);   // wave-uniform
        if (wide) {
            // groups (g, g+1): after the swap lanes < 32 hold cols [g*8, g*8+8), lanes >= 32 hold [(g+1)*8, (g+1)*8+8)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint32_t ax = pk[g][0], ay = pk[g][1], bx = pk[g + 1][0], by = pk[g + 1][1];
                auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                ax = rx[0]; bx = rx[1];
                ay = ry[0]; by = ry[1];
                if (frag_out != nullptr) {   // compile-time known at every call site: the rounded outputs double as MFMA operands
                    u32x4 f = {ax, ay, bx, by};
                    frag_out[g >> 1] = f;
                }
                if (m_ok) {
                    const int co = cbase + (g + hi) * 8;
                    uint16_t* yp;
                    if (a.split > 0 && co >= a.split) yp = reinterpret_cast<uint16_t*>(a.y2) + m * a.y2_cs + (co - a.split);
                    else yp = reinterpret_cast<uint16_t*>(a.y) + m * a.y_cs + co;
                    u32x4 o = {ax, ay, bx, by};
                    *reinterpret_cast<u32x4*>(yp) = o;
                    if (a.up2) {   // the same 8 channels to the 2x2 pixels of the upsampled view (wave-uniform flag)
                        uint16_t* up = reinterpret_cast<uint16_t*>(a.y2) + m_up * a.y2_cs + co;
                        const int64_t row = (int64_t)2 * a.wo * a.y2_cs;
                        *reinterpret_cast<u32x4*>(up) = o;
                        *reinterpret_cast<u32x4*>(up + a.y2_cs) = o;
                        *reinterpret_cast<u32x4*>(up + row) = o;
                        *reinterpret_cast<u32x4*>(up + row + a.y2_cs) = o;
                    }
                }
            }
        } else if (m_ok) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cbase + g * 8 + hi * 4;
                if (co >= a.cout) continue;
                uint16_t* yp;
                if (a.split > 0 && co >= a.split) yp = reinterpret_cast<uint16_t*>(a.y2) + m * a.y2_cs + (co - a.split);
                else yp = reinterpret_cast<uint16_t*>(a.y) + m * a.y_cs + co;
                if (co + 3 < a.cout) {
                    u32x2 o = {pk[g][0], pk[g][1]};
                    *reinterpret_cast<u32x2*>(yp) = o;
                } else {
                    for (int e = 0; e < 4 && co + e < a.cout; ++e) yp[e] = to16<DT>(v[g][e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LEAN epilogue (round 2).  The general finish_subtile above carries every case (narrow couts, a split inside a channel
// octet, the upsampled second output, fp32 output, the chained conv's fragments) behind wave-uniform branches; the code
// was instantiated 4-5 times per kernel (288 v_exp / 324 global_store / 622 branches in a 128x128-tile kernel of 10.7 k
// instructions) and ONE 64x64 wave tile took ~5 k cycles of issue -- five times the MFMA time of a K = 128 1x1 layer.
// This path serves the case every backbone / PAN convolution is in -- 16-bit output, SiLU, full 32-cout sub-tiles, a split
// on a multiple of 16 channels -- with packed fp32 math (v_pk_mul/add_f32), hardware pair conversion
// (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round-to-nearest-even like the scalar path) and one 32-bit per-lane offset per
// pixel group against wave-uniform base pointers.  Same operations in the same order as the general path: bit-identical.
// ---------------------------------------------------------------------------------------------------
// (f32x2 / cvt_pk16: the hardware pair conversion, common.hpp)
template <int DT>
__device__ __forceinline__ f32x2 unpack16(uint32_t u) {
    f32x2 r = {from16<DT>((uint16_t)(u & 0xffffu)), from16<DT>((uint16_t)(u >> 16))};
    return r;
}

// SiLU (+ residual) + rounding + lane swap of one 32x32 sub-tile: o[0] / o[1] are this lane's 16-byte packets of channel
// octets (0 + hi) and (2 + hi) of the sub-tile (lanes < 32: channels [g*8, g*8+8), lanes >= 32: [(g+1)*8, (g+1)*8+8), g = 0, 2)
// -- also exactly the activation fragments of v_mfma_f32_32x32x16 for a chained 1x1 convolution
// PK = false: the same arithmetic with scalar fp32 instructions instead of v_pk_mul_f32 / v_pk_add_f32 (identical results; MI355X_MICROARCH.md prices a packed
// fp32 instruction above two scalar ones when it sits beside MFMAs -- an A/B knob of the fused stem kernel's stage 1)
// SiLU of two fp32 values with packed instructions: x * rcp(1 + exp2(-x * log2 e)) -- THE arithmetic of every 16-bit epilogue (one definition: kernels that
// spread a tile's epilogue over the next tile's MFMAs, conv3x3_res.hip, must round exactly like the ones that run it in one piece)
__device__ __forceinline__ f32x2 silu_pair(f32x2 v) {
    const f32x2 nl2e = {-1.44269504088896341f, -1.44269504088896341f}, one = {1.0f, 1.0f};
    const f32x2 t = v * nl2e;
    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = e + one;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return v * r;
}

template <int DT, bool RES, bool ACT = true, bool PK = true>
__device__ __forceinline__ void silu_pack_subtile(const f32x16& acc, const u32x2 (&rv)[4], u32x4 (&o)[2]) {
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        uint32_t pk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                f32x2 v = {acc[(g + h) * 4 + 2 * p], acc[(g + h) * 4 + 2 * p + 1]};
                if constexpr (ACT && !PK) {
                    float a0 = v[0], a1 = v[1];
                    float e0 = __builtin_amdgcn_exp2f(__fmul_rn(a0, -1.44269504088896341f)), e1 = __builtin_amdgcn_exp2f(__fmul_rn(a1, -1.44269504088896341f));
                    asm volatile("" : "+v"(e0), "+v"(e1));   // keep the two lanes of the pair apart (the SLP vectoriser would re-pack them)
                    e0 = __fadd_rn(e0, 1.0f);
                    e1 = __fadd_rn(e1, 1.0f);
                    asm volatile("" : "+v"(e0), "+v"(e1));
                    const float r0 = __builtin_amdgcn_rcpf(e0), r1 = __builtin_amdgcn_rcpf(e1);
                    a0 = __fmul_rn(a0, r0);
                    a1 = __fmul_rn(a1, r1);
                    asm volatile("" : "+v"(a0), "+v"(a1));
                    v[0] = a0;
                    v[1] = a1;
                } else if constexpr (ACT) {
                    v = silu_pair(v);
                }
                if constexpr (RES) v = v + unpack16<DT>(rv[g + h][p]);
                pk[h][p] = cvt_pk16<DT>(v);
            }
        const auto rx = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        const u32x4 q = {rx[0], ry[0], rx[1], ry[1]};
        o[g >> 1] = q;
    }
}

// 16-byte output store of the lean epilogues.  YMI_NT_STORES (build knob, A/B): nontemporal -- the packet streams towards memory instead of
// sitting dirty in the XCD's L2 until the end-of-kernel write-back
__device__ __forceinline__ void st16(void* p, const u32x4& v) {
#ifdef YMI_NT_STORES
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
#else
    *reinterpret_cast<u32x4*>(p) = v;
#endif
}

// per-lane byte offsets of one pixel group (tensors below 4 GiB: checked by the callers)
struct LeanPix {
    unsigned mo, yo, y2o, ro;   // pixel index (0 when !ok) and its byte offsets in y / y2 / res (+ this lane half's 8- / 4-channel step)
    bool ok;
};
template <class PixFn>
__device__ __forceinline__ LeanPix lean_pix(const ConvArgs& a, int j, int hi, PixFn&& pix) {
    int64_t m;
    LeanPix p;
    pix(j, m, p.ok);
    const unsigned mo = p.ok ? (unsigned)m : 0u;
    p.mo = mo;
    p.yo = (mo * (unsigned)a.y_cs + 8u * (unsigned)hi) * 2u;
    p.y2o = (mo * (unsigned)a.y2_cs + 8u * (unsigned)hi) * 2u;
    p.ro = (mo * (unsigned)a.res_cs + 4u * (unsigned)hi) * 2u;
    if (a.up2) {   // wave-uniform: y2 is the (n, 2ho, 2wo) view; pixel (img, oy, ox) -> top-left of its 2x2 block
        const int hw = a.ho * a.wo;
        const int img = fast_div((int)mo, hw, a.magic_hw);
        const int rem = (int)mo - img * hw;
        const int oy = fast_div(rem, a.wo, a.magic_w), ox = rem - oy * a.wo;
        const unsigned m_up = ((unsigned)(img * 2 * a.ho + 2 * oy) * (unsigned)(2 * a.wo) + 2u * (unsigned)ox);
        p.y2o = (m_up * (unsigned)a.y2_cs + 8u * (unsigned)hi) * 2u;
    }
    return p;
}
// The shortcut arrives in the PACKET form the outputs leave in: lane (pixel, hi) loads the two 16-byte packets of channel octets (0 + hi) and (2 + hi) of each
// 32-channel group -- 32 bytes of 32 pixels per instruction, like a lean store -- and the lane swap of silu_pack_subtile, run backwards, hands every lane the four
// 4-channel pieces of its accumulator rows.  (Round 3: loaded as four 8-byte pieces per group it cost the texture path twice the cache-line look-ups per wave, 256
// instead of 128 for a 64-channel tile, and the persistent 3x3 kernel's timeline showed its waves stalled at exactly these instructions:
// profiles/r03z9_res3x3_timeline.txt.)
__device__ __forceinline__ void unswap_residual_packet(const u32x4& w, u32x2 (&rv)[4], int g) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
    rv[g][0] = s0[0];
    rv[g + 1][0] = s0[1];
    rv[g][1] = s1[0];
    rv[g + 1][1] = s1[1];
}
template <int TN>
__device__ __forceinline__ void lean_load_residual(const ConvArgs& a, int cbase0, const LeanPix& p, u32x2 (&rv)[TN][4], int hi) {
    const char* const rb = reinterpret_cast<const char*>(a.res + cbase0);
    u32x4 w[TN][2] = {};
#pragma unroll
    for (int i = 0; i < TN; ++i)
        {
#pragma unroll
            for (int q = 0; q < 2; ++q)   // wave-uniform: a 16-channel packet pair past cout (cout % 16 == 0, lean_ok) has no shortcut to read
                if (cbase0 + i * 32 + q * 16 < a.cout) w[i][q] = *reinterpret_cast<const u32x4*>(rb + (size_t)p.ro + (size_t)(8 * hi) + (i * 32 + q * 16) * 2);   // p.ro carries 4 * hi channels: + 4 more
        }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) unswap_residual_packet(w[i][q], rv[i], 2 * q);
}
// stores the two packets of sub-tile `cb` (first channel, wave-uniform, a multiple of 32) of one pixel group
__device__ __forceinline__ void lean_store(const ConvArgs& a, const LeanPix& p, int cb, const u32x4 (&o)[2]) {
    if (!p.ok) return;
    char* const yb = reinterpret_cast<char*>(a.y);
    char* const y2b = reinterpret_cast<char*>(a.y2);
    const unsigned up_px = (unsigned)a.y2_cs * 2u, up_row = (unsigned)(2 * a.wo) * (unsigned)a.y2_cs * 2u;   // byte steps of the upsampled view
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int co = cb + q * 16;   // a multiple of 16: the 16 channels of a packet pair are on one side of the split
        if (co >= a.cout) continue;   // ... and on one side of cout (cout % 16 == 0 on this path: lean_ok)
        if (a.split > 0 && co >= a.split) st16(y2b + (size_t)(co - a.split) * 2 + (size_t)p.y2o, o[q]);
        else st16(yb + (size_t)co * 2 + (size_t)p.yo, o[q]);
        if (a.up2) {   // the same 8 channels to the 2x2 pixels of the upsampled view
            char* up = y2b + (size_t)co * 2 + (size_t)p.y2o;
            st16(up, o[q]);
            st16(up + up_px, o[q]);
            st16(up + up_row, o[q]);
            st16(up + up_row + up_px, o[q]);
        }
    }
}

template <int DT, int TN, int TM, bool RES, class PixFn>
__device__ __forceinline__ void finish_wave_tile_lean(const ConvArgs& a, const f32x16 (&acc)[TN][TM], int cbase0, int hi, PixFn&& pix) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const LeanPix p = lean_pix(a, j, hi, pix);
        u32x2 rv[TN][4] = {};
        if constexpr (RES) lean_load_residual<TN>(a, cbase0, p, rv, hi);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if (cbase0 + i * 32 >= a.cout) continue;   // wave-uniform: a whole sub-tile past cout (see lean_ok)
            u32x4 o[2];
            silu_pack_subtile<DT, RES>(acc[i][j], rv[i], o);
            lean_store(a, p, cbase0 + i * 32, o);
        }
    }
}

// wave-uniform preconditions of the lean path; 32-bit byte offsets need every addressed tensor below 4 GiB.
// Every 16-channel packet pair of the wave tile must be either completely inside cout or completely past it: cout % 16 == 0 (round 4: until then the whole wave tile had to
// be inside, and a cout of 96 / 192 / 48 -- yolov5m's widths -- sent every second wave column of a 128-wide block, or the whole 64-wide tile of a 48-cout layer such as its
// stem, through the general epilogue: scalar SiLU, software rounding, per-store predicates).
__device__ __forceinline__ bool lean_ok_whole(const ConvArgs& a, int cbase0, int tn) {
    const int64_t cs_max = a.y_cs > a.y2_cs ? (a.y_cs > a.res_cs ? a.y_cs : a.res_cs) : (a.y2_cs > a.res_cs ? a.y2_cs : a.res_cs);
    return a.act == YMI_ACT_SILU && cbase0 + 32 * tn <= a.cout && (a.split & 15) == 0 && ((int64_t)a.M + 1) * cs_max * (a.up2 ? 4 : 1) < ((int64_t)1 << 31);
}
__device__ __forceinline__ bool lean_ok(const ConvArgs& a, int cbase0, int tn) {
    const int64_t cs_max = a.y_cs > a.y2_cs ? (a.y_cs > a.res_cs ? a.y_cs : a.res_cs) : (a.y2_cs > a.res_cs ? a.y2_cs : a.res_cs);
    return a.act == YMI_ACT_SILU && cbase0 < a.cout && (cbase0 + 32 * tn <= a.cout || (a.cout & 15) == 0) && (a.split & 15) == 0 &&
           ((int64_t)a.M + 1) * cs_max * (a.up2 ? 4 : 1) < ((int64_t)1 << 31);
}

// ROW-TRANSPOSED form of the lean epilogue for wave tiles whose pixel groups are 32 CONSECUTIVE output pixels (the implicit-GEMM
// kernels: lane l of group j holds pixel mbase + 32 j + (l & 31)).  The lean stores write, per instruction, 32 bytes into each of 32
// pixels -- measured 11 % below full-line stores on HBM-streaming layers (tools/partial_line_bench.hip).  Here the packets of a
// pixel group go through a wave-private LDS tile [32 pixels][64 TN + 16 bytes] (the pad keeps the column-wise ds_write_b128
// conflict-free) and leave as whole rows: 64 lanes x 16 bytes = 16 / TN pixels' complete 64 TN-byte channel ranges per instruction.
// Same arithmetic as finish_wave_tile_lean (bit-identical results); `tw` is this wave's 32 * (64 TN + 16) bytes of LDS that nobody
// else touches (the callers put it into the operand ring behind a block barrier).  Opt-in tiles only (141-145, 151-155): written
// at the end of round 2, executed on the CPU simulator (tests/test_hipsim_kernels.py), not yet timed.
template <int TN> constexpr int LEAN_TP_PITCH = 64 * TN + 16;
template <int TN> constexpr int LEAN_TP_BYTES = 32 * LEAN_TP_PITCH<TN>;
// the wave's 32 TN couts go to ONE destination: the channel split not inside the wave's range.  (Round 4: the x2-upsampled copy -- the PAN's nn.Upsample folded into its
// producer -- is written from here too, as whole 64 TN-byte rows to the four pixels of each 2 x 2 block: the lean stores write it as four scattered 32-byte pieces per packet,
// and the two layers that carry it cost 3.5 x / 1.9 x their bare GEMMs, profiles/r04b_gemm_yardstick_c2.txt.)
template <int TN>
__device__ __forceinline__ bool lean_tp_ok(const ConvArgs& a, int cbase0) {
    return lean_ok_whole(a, cbase0, TN) && !(a.split > 0 && cbase0 < a.split && cbase0 + 32 * TN > a.split);
}
template <int DT, int TN, int TM, bool RES>
__device__ __forceinline__ void finish_wave_tile_lean_tp(const ConvArgs& a, const f32x16 (&acc)[TN][TM], int cbase0, int mbase, int lane, unsigned char* tw) {
    static_assert(TN == 1 || TN == 2 || TN == 4, "whole rows per store instruction");
    constexpr int PITCH = LEAN_TP_PITCH<TN>, LPR = 4 * TN, RPI = 64 / LPR;   // lanes per row, rows per store instruction
    const int hi = lane >> 5, frow = lane & 31;
    const bool second = a.split > 0 && cbase0 >= a.split;   // wave-uniform
    char* const yb = second ? reinterpret_cast<char*>(a.y2) + (size_t)(cbase0 - a.split) * 2 : reinterpret_cast<char*>(a.y) + (size_t)cbase0 * 2;
    const size_t ycs = (size_t)(second ? a.y2_cs : a.y_cs) * 2;
    const int row_l = lane / LPR, chunk = lane - row_l * LPR;
    char* const upb = reinterpret_cast<char*>(a.y2) + (size_t)cbase0 * 2;                                  // (a.up2 only) the (n, 2ho, 2wo) view, this wave's couts
    const size_t up_px = (size_t)a.y2_cs * 2, up_row = (size_t)(2 * a.wo) * (size_t)a.y2_cs * 2;            // byte steps of the upsampled view
    const int hw_o = a.ho * a.wo;
    auto pix = [&](int j, int64_t& m, bool& ok) {
        m = mbase + j * 32 + frow;
        ok = m < a.M;
    };
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        u32x2 rv[TN][4] = {};
        if constexpr (RES) {
            const LeanPix p = lean_pix(a, j, hi, pix);
            lean_load_residual<TN>(a, cbase0, p, rv, hi);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            u32x4 o[2];
            silu_pack_subtile<DT, RES>(acc[i][j], rv[i], o);
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(tw + frow * PITCH + (i * 4 + 2 * q + hi) * 16) = o[q];
        }
        __builtin_amdgcn_wave_barrier();   // (scheduling fence: the LDS executes one wave's operations in order)
#pragma unroll
        for (int jj = 0; jj < 2 * TN; ++jj) {
            const int row = jj * RPI + row_l;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tw + row * PITCH + chunk * 16);
            const int64_t mr = (int64_t)mbase + j * 32 + row;
            if (mr < a.M) {
                st16(yb + (size_t)mr * ycs + chunk * 16, v);
                if (a.up2) {   // wave-uniform: the same row to the 2 x 2 pixels of the upsampled view
                    const int img = fast_div((int)mr, hw_o, a.magic_hw);
                    const int rem = (int)mr - img * hw_o;
                    const int oy = fast_div(rem, a.wo, a.magic_w), ox = rem - oy * a.wo;
                    char* up = upb + ((size_t)(img * 2 * a.ho + 2 * oy) * (size_t)(2 * a.wo) + 2 * (size_t)ox) * up_px + chunk * 16;
                    st16(up, v);
                    st16(up + up_px, v);
                    st16(up + up_row, v);
                    st16(up + up_row + up_px, v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// whole wave tile: TM pixel groups x TN cout groups.  pix(j, m, m_ok) yields the output pixel index of this lane in group j.
template <int DT, int ODT, int TN, int TM, class PixFn>
__device__ __forceinline__ void finish_wave_tile(const ConvArgs& a, const f32x16 (&acc)[TN][TM], int cbase0, int hi, PixFn&& pix) {
    if constexpr (ODT == DT) {
        if (lean_ok(a, cbase0, TN)) {
            if (a.res != nullptr) finish_wave_tile_lean<DT, TN, TM, true>(a, acc, cbase0, hi, pix);
            else finish_wave_tile_lean<DT, TN, TM, false>(a, acc, cbase0, hi, pix);
            return;
        }
    }
    int64_t m[TM], m_up[TM];
    bool m_ok[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        pix(j, m[j], m_ok[j]);
        m_up[j] = 0;
        if (a.up2) {   // pixel (img, oy, ox) -> top-left of its 2x2 block in the (n, 2ho, 2wo) view
            const int mm = m_ok[j] ? (int)m[j] : 0;
            const int hw = a.ho * a.wo;
            const int img = fast_div(mm, hw, a.magic_hw);
            const int rem = mm - img * hw;
            const int oy = fast_div(rem, a.wo, a.magic_w), ox = rem - oy * a.wo;
            m_up[j] = ((int64_t)img * 2 * a.ho + 2 * oy) * (2 * a.wo) + 2 * ox;
        }
    }
    if (a.res != nullptr) {   // wave-uniform
        u32x2 rv[TM][TN][4];
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                if (cbase0 + i * 32 < a.cout) load_residual(a, m[j], m_ok[j], cbase0 + i * 32, hi, rv[j][i]);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                if (cbase0 + i * 32 < a.cout) finish_subtile<DT, ODT, true>(a, acc[i][j], m[j], m_ok[j], cbase0 + i * 32, hi, rv[j][i], m_up[j]);
    } else {
        const u32x2 none[4] = {};
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                if (cbase0 + i * 32 < a.cout) finish_subtile<DT, ODT, false>(a, acc[i][j], m[j], m_ok[j], cbase0 + i * 32, hi, none, m_up[j]);
    }
}

// Wave tile with a CHAINED 1x1 convolution (ymi_conv_desc.chain_w): the wave owns pixel groups j and ALL chain_k = 32*TN
// output channels of the first conv; its rounded, lane-swapped 16-byte output packets are exactly the activation
// fragments of v_mfma_f32_32x32x16 (lanes < 32: channels 16s..16s+7, lanes >= 32: 16s+8..16s+15 of pixel lane & 31), so
// the second GEMM runs from registers; its weight fragments come straight from global memory (L2 resident, <= 16 KB).
template <int DT, int TN, int TM, class PixFn>
__device__ __forceinline__ void finish_wave_tile_chain(const ConvArgs& a, const f32x16 (&acc)[TN][TM], int hi, int lane, PixFn&& pix) {
    typedef typename Mfma<DT>::frag frag;
    constexpr int S2MAX = 8;   // second-source k16 steps (chain_k2 <= 128)
    // lean form (the case every chained launch of the YOLOv5 graphs is in): one K source, no shortcut on the producer
    if (a.res == nullptr && a.chain_x2 == nullptr && !a.up2 && lean_ok_whole(a, 0, TN) && ((int64_t)a.M + 1) * a.chain_y_cs < ((int64_t)1 << 31)) {
        ConvArgs a2 = a;   // output side of the chained conv
        a2.y = a.chain_y; a2.y_cs = a.chain_y_cs; a2.split = 0; a2.up2 = 0;
        const int tn2 = a.chain_cout >> 5;   // 1..4 (wave-uniform)
        const u32x2 none[4] = {};
        LeanPix px[TM];
        u32x4 fr[TM][TN][2];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            px[j] = lean_pix(a, j, hi, pix);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                silu_pack_subtile<DT, false>(acc[i][j], none, fr[j][i]);
                lean_store(a, px[j], i * 32, fr[j][i]);
            }
            px[j].yo = (px[j].mo * (unsigned)a.chain_y_cs + 8u * (unsigned)hi) * 2u;   // the same pixel in chain_y (used with a2 below)
        }
        for (int i2 = 0; i2 < tn2; ++i2) {
            frag wf[2 * TN];
            const uint16_t* wr = a.chain_w + (int64_t)(i2 * 32 + (lane & 31)) * a.chain_k + 8 * hi;
#pragma unroll
            for (int s2 = 0; s2 < 2 * TN; ++s2) wf[s2] = *reinterpret_cast<const frag*>(wr + 16 * s2);
            f32x4 b2[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b2[g] = *reinterpret_cast<const f32x4*>(a.chain_bias + i2 * 32 + g * 8 + hi * 4);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                f32x16 acc2;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[g * 4 + e] = b2[g][e];
#pragma unroll
                for (int s2 = 0; s2 < 2 * TN; ++s2) {
                    frag xf;
                    __builtin_memcpy(&xf, &fr[j][s2 >> 1][s2 & 1], 16);
                    acc2 = Mfma<DT>::run(wf[s2], xf, acc2);
                }
                u32x4 o[2];
                silu_pack_subtile<DT, false>(acc2, none, o);
                lean_store(a2, px[j], i2 * 32, o);
            }
        }
        return;
    }
    int64_t m[TM];
    bool m_ok[TM];
    u32x4 fr[TM][TN][2];
    const u32x2 none[4] = {};
#pragma unroll
    for (int j = 0; j < TM; ++j) pix(j, m[j], m_ok[j]);
    if (a.res != nullptr) {   // wave-uniform: the producer is a Bottleneck.cv2 with its shortcut
        u32x2 rv[TM][TN][4];
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) load_residual(a, m[j], m_ok[j], i * 32, hi, rv[j][i]);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) finish_subtile<DT, DT, true>(a, acc[i][j], m[j], m_ok[j], i * 32, hi, rv[j][i], 0, fr[j][i]);
    } else {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) finish_subtile<DT, DT, false>(a, acc[i][j], m[j], m_ok[j], i * 32, hi, none, 0, fr[j][i]);
    }
    // second K range: 8 channels per lane and k16 step straight from the other tensor (zeros for pixels past the end)
    const int s2n = a.chain_x2 != nullptr ? a.chain_k2 >> 4 : 0;   // wave-uniform
    u32x4 xr[TM][S2MAX];
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int s2 = 0; s2 < S2MAX; ++s2) {
            u32x4 z = {0u, 0u, 0u, 0u};
            xr[j][s2] = z;
            if (s2 < s2n && m_ok[j]) xr[j][s2] = *reinterpret_cast<const u32x4*>(a.chain_x2 + m[j] * a.chain_x2_cs + 16 * s2 + 8 * hi);
        }
    ConvArgs a2 = a;   // output side of the chained conv
    a2.y = a.chain_y; a2.y_cs = a.chain_y_cs; a2.cout = a.chain_cout; a2.cout_pad = a.chain_cout; a2.split = 0; a2.up2 = 0; a2.res = nullptr;
    const int ktot = a.chain_k + (a.chain_x2 != nullptr ? a.chain_k2 : 0);
    const int tn2 = a.chain_cout >> 5;   // 1..4 (wave-uniform)
    for (int i2 = 0; i2 < tn2; ++i2) {
        // weight fragments of cout rows i2*32 + (lane & 31): 2*TN k16-steps for the fresh outputs, then the second source's
        frag wf[2 * TN + S2MAX];
        const uint16_t* wr = a.chain_w + (int64_t)(i2 * 32 + (lane & 31)) * ktot + 8 * hi;
#pragma unroll
        for (int s2 = 0; s2 < 2 * TN + S2MAX; ++s2)
            if (s2 < 2 * TN + s2n) wf[s2] = *reinterpret_cast<const frag*>(wr + 16 * s2);
        f32x4 b2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b2[g] = *reinterpret_cast<const f32x4*>(a.chain_bias + i2 * 32 + g * 8 + hi * 4);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            f32x16 acc2;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[g * 4 + e] = b2[g][e];
#pragma unroll
            for (int s2 = 0; s2 < 2 * TN; ++s2) {
                frag xf;
                __builtin_memcpy(&xf, &fr[j][s2 >> 1][s2 & 1], 16);
                acc2 = Mfma<DT>::run(wf[s2], xf, acc2);
            }
#pragma unroll
            for (int s2 = 0; s2 < S2MAX; ++s2)
                if (s2 < s2n) {
                    frag xf;
                    __builtin_memcpy(&xf, &xr[j][s2], 16);
                    acc2 = Mfma<DT>::run(wf[2 * TN + s2], xf, acc2);
                }
            finish_subtile<DT, DT, false>(a2, acc2, m[j], m_ok[j], i2 * 32, hi, none);
        }
    }
}

// fp32 parity-mode kernel (conv_f32.hip): fp32 activations / weights / arithmetic
int conv_f32_launch(const ConvArgs& a, bool is1x1, hipStream_t s);
// fp32 mode, LDS-DMA pipelined tiles 201-206 (conv_f32_pipe.hip; tile 0 = chosen from the shape by conv_f32_pick_tile)
int conv_f32_pipe_launch(const ConvArgs& a, bool is1x1, int tile, hipStream_t s);
int conv_f32_pick_tile(int M, int cout_pad);
// 3x3 stride-1 LDS-halo kernel (conv3x3_halo.hip); returns YMI_EINVAL when the shape does not apply
int conv3x3_halo_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// 8-wave LDS-halo 3x3 stride-1 kernel (conv_halo8.hip), patch shape chosen per feature-map size
int conv_halo8_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// 8-wave implicit-GEMM kernel with 64-deep steps (conv_igemm8.hip): 1x1 / strided k x k layers with cin % 32 == 0
int conv_igemm8_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// resident-weights persistent 3x3 kernel for cin = 32, cout = 32 / 64, stride 1 / 2 (conv3x3_c32.hip)
int conv3x3_c32_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// ... for cin = 48 / 64, cout <= 64, stride 1, cross-tile patch prefetch (conv3x3_res.hip)
int conv3x3_res_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// ... weights in registers, two independent 4-wave blocks per CU (conv3x3_rw.hip; opt-in)
int conv3x3_rw_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// ... the same at stride 2, cin = 64 -> cout = 128 (conv3x3_rw2.hip, tile 134)
int conv3x3_rw2_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// streaming 1x1 kernel for the memory-bound shallow-K layers (conv1x1_stream.hip): no LDS, weights in registers, variant = cout tiles per wave
int conv1x1_stream_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);
// dedicated stem kernel (conv_stem.hip): 6x3 s(2,1) super-pixel form, input patch in LDS, weights in registers
int conv_stem_launch(const ConvArgs& a, int dtype, int out_dtype, hipStream_t s);
// the same stem fed from planar (3, H, W) images of the compute dtype, identity-size batches (no letterbox pass)
int conv_stem_planar_launch(const ConvArgs& a, const void* const* imgs, int dtype, int out_dtype, hipStream_t s);
int stem_body1_planar_launch(const ConvArgs& a1, const ConvArgs& a2, const void* const* imgs, int dtype, hipStream_t s);   // stem_body1_fused.hip
int stem_body1_launch(const ConvArgs& a1, const ConvArgs& a2, int dtype, hipStream_t s);                                    // stem_body1_fused.hip (NHWC4 canvas)

}  // namespace ymi
