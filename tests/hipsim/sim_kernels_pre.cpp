// fourth translation unit of the simulator build (see sim_kernels.cpp): csrc/preproc_pool.hip -- the letterbox kernels (per-pixel, tiled,
// identity copy), the SPP max-pool cascade, nearest x2 upsample, view copy and the NCHW <-> NHWC edges.  Its extern "C" entry points
// (ymi_letterbox, ymi_spp_pool, ymi_upsample2x, ymi_copy_view, ymi_nchw_to_nhwc, ymi_nhwc_to_nchw) are exported as they are: in this
// library they take HOST pointers.
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
constexpr int SIM_LDS = 160 * 1024;
alignas(16) unsigned char lb_sm[SIM_LDS];
alignas(16) u32x4 spp_sm[SIM_LDS / 16];
alignas(16) unsigned char spp32_sm[SIM_LDS];
}

#include "../../yolort_amd/csrc/preproc_pool.hip"
