// translation unit of the simulator build (see sim_kernels.cpp): the strip kernel of a whole C3 / its Bottlenecks (csrc/c3_tile.hip) and its weight packer
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
constexpr int SIM_LDS = 160 * 1024;
alignas(16) unsigned char c3t_sm[SIM_LDS];
}

#include "../../yolort_amd/csrc/c3_tile.hip"

extern "C" int sim_c3_tile(const ymi_c3_desc* d) { return ymi::c3_tile_launch(d, nullptr); }
