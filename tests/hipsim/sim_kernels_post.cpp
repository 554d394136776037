// fifth translation unit of the simulator build (see sim_kernels.cpp): csrc/postprocess.hip -- decode, threshold, per-image ranking
// sort, class-aware NMS (wave64 ballot / shuffle), top-k and rescale.  Built from a COPY of the source in which `extern __shared__`
// lost the keyword and with -D__shared__=static (static __shared__ arrays become function-local statics: one block runs at a time).
// Exports ymi_batched_nms / ymi_postprocess / ... as they are; in this library they take HOST pointers.
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
constexpr int SIM_LDS = 160 * 1024;
alignas(16) uint64_t lds_key[SIM_LDS / 8];
}

#include "postprocess.sim.hip"
