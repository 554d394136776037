// second translation unit of the simulator build (see sim_kernels.cpp): the tiled convolution kernels
//   tiles 91-95  conv_halo8_kernel   111-116 conv_igemm8_kernel   31-37 conv3x3_halo_kernel   (12/21/24/27/61/64/66 conv_igemm_v2_kernel: sim_kernels_v2.cpp)
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
constexpr int SIM_LDS = 160 * 1024;
extern uint16_t smem[SIM_LDS / 2];   // defined in sim_kernels.cpp
}

#include "../../yolort_amd/csrc/conv_halo8.hip"
#include "../../yolort_amd/csrc/conv_igemm8.hip"
#include "../../yolort_amd/csrc/conv3x3_halo.hip"

int sim_conv2d_v2(const ymi::ConvArgs& a, const ymi_conv_desc* d);   // sim_kernels_v2.cpp

int sim_conv2d_gemm(const ymi::ConvArgs& a, const ymi_conv_desc* d) {
    if (d->tile >= 91 && d->tile <= 99) return ymi::conv_halo8_launch(a, d->dtype, d->out_dtype, d->tile - 90, nullptr);
    if (d->tile >= 111 && d->tile <= 120) return ymi::conv_igemm8_launch(a, d->dtype, d->out_dtype, d->tile - 110, nullptr);
    if (d->tile >= 151 && d->tile <= 159) return ymi::conv_igemm8_launch(a, d->dtype, d->out_dtype, d->tile - 140, nullptr);   // row-transposed stores
    if (d->tile >= 31 && d->tile <= 39) return ymi::conv3x3_halo_launch(a, d->dtype, d->out_dtype, d->tile - 30, nullptr);
    return sim_conv2d_v2(a, d);
}
