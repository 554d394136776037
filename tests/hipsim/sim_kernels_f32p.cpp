// translation unit of the simulator build (see sim_kernels.cpp): csrc/conv_f32_pipe.hip -- the LDS-DMA pipelined fp32 convolution (tiles 201-206; round 5)
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
alignas(16) unsigned char f32p_sm[160 * 1024];   // the kernel's `extern __shared__` array: one CU's worth of LDS (blocks run one after another)
}

#include "../../yolort_amd/csrc/conv_f32_pipe.hip"

int sim_conv2d_f32p(const ymi::ConvArgs& a, const ymi_conv_desc* d) {
    const bool is1x1 = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0;
    if (d->out_dtype != YMI_F32) { ymi::set_error("sim_conv2d: the fp32 kernels store fp32"); return YMI_EINVAL; }
    return ymi::conv_f32_pipe_launch(a, is1x1, d->tile, nullptr);
}
extern "C" int sim_conv_f32_pick_tile(int M, int cout_pad) { return ymi::conv_f32_pick_tile(M, cout_pad); }
