// seventh translation unit of the simulator build (see sim_kernels.cpp): the detection head with the decode fused into its epilogue
// (csrc/head_decode.hpp, conv_head_decode_group_kernel of csrc/conv_igemm_impl.hpp) for K = 85 outputs per anchor (TNA = 3), anchor-split
// form.  The argument construction restates head_decode_prepare / conv_head_decode_group_launch of csrc/conv_igemm.hip (that file drags
// the whole tile space in); ymi_post_begin / ymi_post_finish come from the post-process unit.
#include "hipsim.h"

#include "../../yolort_amd/csrc/common.hpp"

namespace ymi {
constexpr int SIM_LDS = 160 * 1024;
extern uint16_t smem[SIM_LDS / 2];   // defined in sim_kernels.cpp
}

#include "../../yolort_amd/csrc/conv_igemm_impl.hpp"
#include "sim_fill.h"

extern "C" int sim_conv_head_decode_group(const ymi_conv_desc* descs, int n_levels, const ymi_post_desc* post) {
    using namespace ymi;
    YMI_REQUIRE(descs && post && n_levels >= 1 && n_levels <= YMI_MAX_LEVELS && n_levels == post->num_levels, "sim_conv_head_decode_group: bad arguments");
    const int K = post->num_classes + 5, ra = (K + 31) / 32 * 32;
    YMI_REQUIRE(ra == 96, "sim_conv_head_decode_group: the simulator build holds the 85-outputs-per-anchor instance");
    HeadGroupArgs g;
    memset(&g, 0, sizeof(g));
    g.n = n_levels;
    const PostLayout L = post_layout(post);
    const Workspace w = carve(post->ws, post->n, L.total_anchors, post->cand_cap);
    YMI_REQUIRE(post->ws_bytes >= w.total, "sim_conv_head_decode_group: workspace too small");
    for (int l = 0; l < n_levels; ++l) {
        const ymi_conv_desc* d = &descs[l];
        YMI_REQUIRE(d->cout == 3 * ra && d->cout_pad == 3 * ra && d->k_pad == d->cin && d->zeros, "sim_conv_head_decode_group: anchor-padded packing expected");
        sim_fill(d, g.a[l]);
        g.a[l].nblk_m = cdiv(g.a[l].M, 128);
        g.a[l].nblk_n = 3;   // anchor split (the shipped default)
        HeadDecodeArgs& h = g.h[l];
        h.stride = post->stride[l];
        for (int k = 0; k < 6; ++k) h.anc[k] = post->anchors[l][k];
        h.K = K;
        h.level_off = 0;
        for (int q = 0; q < l; ++q) h.level_off += 3 * post->lh[q] * post->lw[q];
        h.sink = make_sink(post, w, L);
    }
    int blocks = 0;
    for (int l = n_levels - 1; l >= 0; --l) {
        g.first_block[l] = blocks;
        blocks += g.a[l].nblk_m * g.a[l].nblk_n;
    }
    g.first_block[n_levels] = blocks;
    if (blocks == 0) return YMI_OK;
    return descs[0].dtype == YMI_F16 ? launch_head_group<YMI_F16, 3>(g, nullptr) : launch_head_group<YMI_BF16, 3>(g, nullptr);
}
