// translation unit of the simulator build (see sim_kernels.cpp): csrc/conv_f32.hip -- the fp32 PARITY MODE convolution (exact fp32 arithmetic
// on v_mfma_f32_32x32x2_f32, blocked summation).  Built with -D__shared__=static (static LDS arrays).
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"
#include "../../yolort_amd/csrc/conv_f32.hip"

int sim_conv2d_f32(const ymi::ConvArgs& a, const ymi_conv_desc* d) {
    const bool is1x1 = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0;
    if (d->out_dtype != YMI_F32) { ymi::set_error("sim_conv2d: the fp32 parity kernel stores fp32"); return YMI_EINVAL; }
    return ymi::conv_f32_launch(a, is1x1, nullptr);
}
