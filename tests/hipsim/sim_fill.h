// descriptor -> kernel arguments, shared by the simulator's translation units
#pragma once
// descriptor -> kernel arguments: the fields conv_igemm.hip's fill_conv_args sets (that file is not part of this build)
inline void sim_fill(const ymi_conv_desc* d, ymi::ConvArgs& a) {
    memset(&a, 0, sizeof(a));
    a.x = (const uint16_t*)d->x; a.w = (const uint16_t*)d->w; a.bias = d->bias; a.ktab = (const int2*)d->ktab;
    a.y = d->y; a.res = (const uint16_t*)d->res;
    a.n = d->n; a.h = d->h; a.w_in = d->w_in; a.cin = d->cin; a.x_cs = d->x_cstride;
    a.ho = d->ho; a.wo = d->wo; a.cout = d->cout; a.cout_pad = d->cout_pad; a.y_cs = d->y_cstride; a.res_cs = d->res_cstride;
    a.sh = d->sh; a.sw = d->sw; a.ph = d->ph; a.pw = d->pw; a.k_pad = d->k_pad; a.act = d->act;
    a.M = d->n * d->ho * d->wo;
    a.y2 = d->y2; a.y2_cs = d->y2_cstride; a.split = d->cout_split; a.zeros = (const uint16_t*)d->zeros;
    a.up2 = d->y2_mode == 1 ? 1 : 0;
    a.chain_w = (const uint16_t*)d->chain_w; a.chain_bias = d->chain_bias; a.chain_y = d->chain_y;
    a.chain_cout = d->chain_cout; a.chain_y_cs = d->chain_y_cstride; a.chain_k = d->cout_split > 0 ? d->cout_split : d->cout;
    a.chain_x2 = (const uint16_t*)d->chain_x2; a.chain_x2_cs = d->chain_x2_cstride; a.chain_k2 = d->chain_x2 != nullptr ? d->chain_k2 : 0;
    a.kh = d->kh; a.kw = d->kw;
    auto magic = [](int dv) { const uint64_t v = (((uint64_t)1 << 32) / (uint64_t)dv) + 1u; return (unsigned)(v > 0xffffffffull ? 0xffffffffull : v); };
    a.magic_hw = magic(d->ho * d->wo);
    a.magic_w = magic(d->wo);
    if (d->zeros != nullptr) a.x_zero_off = (int)(((const char*)d->zeros - (const char*)d->x) / 2);
}
