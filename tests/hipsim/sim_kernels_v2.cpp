// third translation unit of the simulator build (see sim_kernels.cpp): LDS-DMA implicit-GEMM tiles 12/13/21/24/26/27/61/64/66 (+ 141-144)
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
constexpr int SIM_LDS = 160 * 1024;
extern uint16_t smem[SIM_LDS / 2];   // defined in sim_kernels.cpp
}

#include "../../yolort_amd/csrc/conv_igemm_impl.hpp"

namespace {
// the LDS-DMA implicit-GEMM tiles of the yolov5s table (csrc/conv_igemm_impl.hpp; the product spreads the whole tile space over
// 16 translation units, here only these are instantiated)
template <int DT>
int sim_v2(const ymi::ConvArgs& a, bool is1x1, int tile) {
    using namespace ymi;
    switch (tile) {
        case 12: return launch_v2<DT, DT, 256, 64, 64, 64, 3>(a, is1x1, nullptr);
        case 13: return launch_v2<DT, DT, 256, 32, 64, 32, 3>(a, is1x1, nullptr);
        case 21: return launch_v2<DT, DT, 128, 128, 64, 64, 2>(a, is1x1, nullptr);
        case 26: return launch_v2<DT, DT, 128, 32, 32, 32, 2>(a, is1x1, nullptr);
        case 24: return launch_v2<DT, DT, 64, 128, 32, 64, 2>(a, is1x1, nullptr);
        case 27: return launch_v2<DT, DT, 64, 64, 32, 32, 2>(a, is1x1, nullptr);
        case 61: return launch_v2<DT, DT, 128, 128, 64, 64, 4, true>(a, is1x1, nullptr);
        case 64: return launch_v2<DT, DT, 64, 128, 32, 64, 4, true>(a, is1x1, nullptr);
        case 66: return launch_v2<DT, DT, 256, 128, 128, 64, 3, true>(a, is1x1, nullptr);
        // the same tiles with row-transposed stores (StoreEpilogueTP)
        case 141: return launch_v2<DT, DT, 256, 64, 64, 64, 3, false, true>(a, is1x1, nullptr);
        case 142: return launch_v2<DT, DT, 128, 128, 64, 64, 2, false, true>(a, is1x1, nullptr);
        case 143: return launch_v2<DT, DT, 256, 128, 128, 64, 3, true, true>(a, is1x1, nullptr);
        case 144: return launch_v2<DT, DT, 128, 128, 64, 64, 4, true, true>(a, is1x1, nullptr);
        default: break;
    }
    set_error("sim_conv2d: implicit-GEMM tile %d is not instantiated in the simulator build", tile);
    return YMI_EINVAL;
}
}  // namespace


int sim_conv2d_v2(const ymi::ConvArgs& a, const ymi_conv_desc* d) {
    const bool is1x1 = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0;
    if (d->out_dtype == YMI_F32 && d->tile == 21) {   // the unfused head: fp32 logits (box_head.py:74)
        return d->dtype == YMI_F16 ? ymi::launch_v2<YMI_F16, YMI_F32, 128, 128, 64, 64, 2>(a, is1x1, nullptr) : ymi::launch_v2<YMI_BF16, YMI_F32, 128, 128, 64, 64, 2>(a, is1x1, nullptr);
    }
    if (d->out_dtype != d->dtype) { ymi::set_error("sim_conv2d: fp32 outputs through tile 21 only"); return YMI_EINVAL; }
    return d->dtype == YMI_F16 ? sim_v2<YMI_F16>(a, is1x1, d->tile) : sim_v2<YMI_BF16>(a, is1x1, d->tile);
}
