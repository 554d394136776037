// sixth translation unit of the simulator build (see sim_kernels.cpp): csrc/conv_stem.hip -- the 6x6 stride-2 stem in its super-pixel form
// from the NHWC4 batch (tile 41) and straight from planar images (ymi_conv_stem_planar).  Built with -D__shared__=static (static LDS arrays).
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"
#include "../../yolort_amd/csrc/conv_stem.hip"
#include "sim_fill.h"

int sim_conv2d_stem(const ymi::ConvArgs& a, const ymi_conv_desc* d) { return ymi::conv_stem_launch(a, d->dtype, d->out_dtype, nullptr); }

extern "C" int sim_conv_stem_planar(const ymi_conv_desc* d, const void* const* imgs, int n_imgs) {
    ymi::ConvArgs a;
    sim_fill(d, a);
    if (n_imgs != d->n) { ymi::set_error("sim_conv_stem_planar: %d images for a batch of %d", n_imgs, d->n); return YMI_EINVAL; }
    return ymi::conv_stem_planar_launch(a, imgs, d->dtype, d->out_dtype, nullptr);
}

