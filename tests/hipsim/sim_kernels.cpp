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
alignas(16) unsigned char sb_sm[SIM_LDS];
alignas(16) unsigned char r3_sm[SIM_LDS];
alignas(16) unsigned char rw_sm[SIM_LDS];
alignas(16) unsigned char r2_sm[SIM_LDS];
alignas(16) unsigned char r5_sm[SIM_LDS];
alignas(16) unsigned char rs_sm[SIM_LDS];
alignas(16) uint16_t smem[SIM_LDS / 2];
}  // namespace ymi

#include "../../yolort_amd/csrc/c3_fused32.hip"
#include "../../yolort_amd/csrc/conv1x1_stream.hip"
#include "../../yolort_amd/csrc/conv3x3_c32.hip"
#include "../../yolort_amd/csrc/conv3x3_res.hip"
#include "../../yolort_amd/csrc/conv3x3_rw.hip"
#include "../../yolort_amd/csrc/conv3x3_rw2.hip"
#include "../../yolort_amd/csrc/conv3x3_rs.hip"
#include "../../yolort_amd/csrc/stem_body1_fused.hip"

int sim_conv2d_gemm(const ymi::ConvArgs& a, const ymi_conv_desc* d);
int sim_conv2d_stem(const ymi::ConvArgs& a, const ymi_conv_desc* d);   // sim_kernels_stem.cpp
int sim_conv2d_f32(const ymi::ConvArgs& a, const ymi_conv_desc* d);    // sim_kernels_f32.cpp
int sim_conv2d_f32p(const ymi::ConvArgs& a, const ymi_conv_desc* d);   // sim_kernels_f32p.cpp

#include "sim_fill.h"

extern "C" int sim_c3_fused(const ymi_c3_desc* d) { return ymi_c3_fused(d, nullptr); }
extern "C" int sim_conv2d(const ymi_conv_desc* d) {
    ymi::ConvArgs a;
    sim_fill(d, a);
    if (d->dtype == YMI_F32) return (d->tile >= 0 && d->zeros != nullptr) ? sim_conv2d_f32p(a, d) : sim_conv2d_f32(a, d);   // fp32 mode: pipelined tiles 201-206 (0 = by shape) / the register-staged kernel (negative tile)
    if (d->tile >= 121 && d->tile <= 124) return ymi::conv1x1_stream_launch(a, d->dtype, d->out_dtype, d->tile - 120, nullptr);
    if (d->tile == 131) return ymi::conv3x3_c32_launch(a, d->dtype, d->out_dtype, 1, nullptr);
    if (d->tile == 132) return ymi::conv3x3_res_launch(a, d->dtype, d->out_dtype, 1, nullptr);
    if (d->tile == 133) return ymi::conv3x3_rw_launch(a, d->dtype, d->out_dtype, 1, nullptr);
    if (d->tile == 134) return ymi::conv3x3_rw2_launch(a, d->dtype, d->out_dtype, 1, nullptr);
    if (d->tile == 135) return ymi::conv3x3_rw2_launch(a, d->dtype, d->out_dtype, 2, nullptr);
    if (d->tile == 137 || d->tile == 138) return ymi::conv3x3_rs_launch(a, d->dtype, d->out_dtype, d->tile - 136, nullptr);
    if (d->tile == 41) return sim_conv2d_stem(a, d);
    if ((d->tile >= 11 && d->tile <= 120) || (d->tile >= 141 && d->tile <= 159)) return sim_conv2d_gemm(a, d);   // sim_kernels_gemm.cpp
    ymi::set_error("sim_conv2d: tile %d is not part of the simulator build", d->tile);
    return YMI_EINVAL;
}
// stem + body.1 in one launch, from planar images (csrc/stem_body1_fused.hip)
extern "C" int sim_stem_body1_planar(const ymi_conv_desc* stem, const ymi_conv_desc* body1, const void* const* imgs, int n_imgs) {
    ymi::ConvArgs a1, a2;
    sim_fill(stem, a1);
    sim_fill(body1, a2);
    if (n_imgs != stem->n || n_imgs != body1->n) { ymi::set_error("sim_stem_body1_planar: %d images for batches of %d / %d", n_imgs, stem->n, body1->n); return YMI_EINVAL; }
    return ymi::stem_body1_planar_launch(a1, a2, imgs, stem->dtype, nullptr);
}
// the same from the letterboxed NHWC4 canvas (stem->x)
extern "C" int sim_stem_body1(const ymi_conv_desc* stem, const ymi_conv_desc* body1) {
    ymi::ConvArgs a1, a2;
    sim_fill(stem, a1);
    sim_fill(body1, a2);
    return ymi::stem_body1_launch(a1, a2, stem->dtype, nullptr);
}
extern "C" const char* sim_last_error(void) { return ymi::g_err; }
extern "C" int sim_max_lds(void) { return hipsim::g_max_lds; }
extern "C" long sim_launches(void) { return hipsim::g_launches; }
