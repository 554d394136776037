// CPU build of selected kernels of yolort_amd/csrc on the hipsim runtime (TEST INFRASTRUCTURE: tests/test_hipsim_kernels.py builds
// this file with the host clang++ and drives it through ctypes; nothing here is part of libyolort_amd.so).
//   sim_c3_fused   -> ymi_c3_fused                        (csrc/c3_fused32.hip)
//   sim_conv2d     -> conv1x1_stream_launch (tiles 121-124) / conv3x3_c32_launch (tile 131), the launches the fused C3 replaces
//                   (+ sim_kernels_gemm.cpp, a second translation unit so that the two compile in parallel: the 8-wave and 4-wave
//                   LDS-halo 3x3 kernels, the 8-wave implicit GEMM and a few LDS-DMA implicit-GEMM tiles)
// All pointers are host memory.
#include "hipsim.h"
#include "hipsim.cpp"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
static char g_err[512];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int allow_big_lds(const void*, int) { return YMI_OK; }
size_t lds_floor_bytes() { return 0; }
bool head_anchor_split() { return true; }
// the kernels' `extern __shared__` arrays: one CU's worth of LDS each (blocks run one after another)
constexpr int SIM_LDS = 160 * 1024;
alignas(16) unsigned char f3_sm[SIM_LDS];
alignas(16) unsigned char st_sm[SIM_LDS];
alignas(16) unsigned char c32_sm[SIM_LDS];
alignas(16) uint16_t smem[SIM_LDS / 2];
}  // namespace ymi

#include "../../yolort_amd/csrc/c3_fused32.hip"
#include "../../yolort_amd/csrc/conv1x1_stream.hip"
#include "../../yolort_amd/csrc/conv3x3_c32.hip"

int sim_conv2d_gemm(const ymi::ConvArgs& a, const ymi_conv_desc* d);

namespace {
// descriptor -> kernel arguments: the fields conv_igemm.hip's fill_conv_args sets (that file is not part of this build)
void sim_fill(const ymi_conv_desc* d, ymi::ConvArgs& a) {
    memset(&a, 0, sizeof(a));
    a.x = (const uint16_t*)d->x; a.w = (const uint16_t*)d->w; a.bias = d->bias; a.ktab = (const int2*)d->ktab;
    a.y = d->y; a.res = (const uint16_t*)d->res;
    a.n = d->n; a.h = d->h; a.w_in = d->w_in; a.cin = d->cin; a.x_cs = d->x_cstride;
    a.ho = d->ho; a.wo = d->wo; a.cout = d->cout; a.cout_pad = d->cout_pad; a.y_cs = d->y_cstride; a.res_cs = d->res_cstride;
    a.sh = d->sh; a.sw = d->sw; a.ph = d->ph; a.pw = d->pw; a.k_pad = d->k_pad; a.act = d->act;
    a.M = d->n * d->ho * d->wo;
    a.y2 = d->y2; a.y2_cs = d->y2_cstride; a.split = d->cout_split; a.zeros = (const uint16_t*)d->zeros;
    a.up2 = d->y2_mode == 1 ? 1 : 0;
    a.chain_w = (const uint16_t*)d->chain_w; a.chain_bias = d->chain_bias; a.chain_y = d->chain_y;
    a.chain_cout = d->chain_cout; a.chain_y_cs = d->chain_y_cstride; a.chain_k = d->cout_split > 0 ? d->cout_split : d->cout;
    a.chain_x2 = (const uint16_t*)d->chain_x2; a.chain_x2_cs = d->chain_x2_cstride; a.chain_k2 = d->chain_x2 != nullptr ? d->chain_k2 : 0;
    a.kh = d->kh; a.kw = d->kw;
    auto magic = [](int dv) { const uint64_t v = (((uint64_t)1 << 32) / (uint64_t)dv) + 1u; return (unsigned)(v > 0xffffffffull ? 0xffffffffull : v); };
    a.magic_hw = magic(d->ho * d->wo);
    a.magic_w = magic(d->wo);
    if (d->zeros != nullptr) a.x_zero_off = (int)(((const char*)d->zeros - (const char*)d->x) / 2);
}
}  // namespace

extern "C" int sim_c3_fused(const ymi_c3_desc* d) { return ymi_c3_fused(d, nullptr); }
extern "C" int sim_conv2d(const ymi_conv_desc* d) {
    ymi::ConvArgs a;
    sim_fill(d, a);
    if (d->tile >= 121 && d->tile <= 124) return ymi::conv1x1_stream_launch(a, d->dtype, d->out_dtype, d->tile - 120, nullptr);
    if (d->tile == 131) return ymi::conv3x3_c32_launch(a, d->dtype, d->out_dtype, 1, nullptr);
    if ((d->tile >= 11 && d->tile <= 119) || (d->tile >= 141 && d->tile <= 159)) return sim_conv2d_gemm(a, d);   // sim_kernels_gemm.cpp
    ymi::set_error("sim_conv2d: tile %d is not part of the simulator build", d->tile);
    return YMI_EINVAL;
}
extern "C" const char* sim_last_error(void) { return ymi::g_err; }
extern "C" int sim_max_lds(void) { return hipsim::g_max_lds; }
extern "C" long sim_launches(void) { return hipsim::g_launches; }
