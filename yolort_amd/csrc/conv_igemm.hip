// Implicit-GEMM convolution for gfx950 (MI355X): NHWC fp16/bf16 activations, MFMA 32x32x16,
// fused bias + SiLU (+ residual) epilogue, reads/writes channel slices of concat buffers.
//
// GEMM view (computed "swapped" so that each lane ends up owning consecutive output channels of
// ONE pixel, which makes the NHWC store 8/16 bytes wide):
//     D[cout][pixel] = sum_k  W[cout][k] * X[pixel][k],   k = (ky*kw + kx)*cin + c
//   MFMA A operand = weight fragment (rows = cout), B operand = activation fragment (cols = pixel).
//   v_mfma_f32_32x32x16: lane l holds A[row = l&31][k = 8*(l>>5) .. +7], B[k = 8*(l>>5)..+7][col = l&31],
//   D reg r of lane l = D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
//
// Tiling: block = 256 threads = 4 waves, block tile BM pixels x BN couts, BK = 32 per step.
// Activation tile is gathered straight from the NHWC input with 16-byte loads (8 channels of one
// tap of one pixel), zero-filled outside the image; an int32 table (ktab) gives, per 8-channel
// k-chunk, the element offset and (dy,dx) of its tap, so 1x1, 3x3 s1/s2 and the 6x6 s2 stem share
// one kernel.  Tiles are double-buffered in LDS (register-staged: global loads for step t+1 are
// issued before the MFMAs of step t; the LDS write lands after them).  LDS rows are padded to
// 80 bytes: a 16-lane ds_read_b128 group then touches 16 distinct 16-byte slots (conflict free).
//
// Replaces: yolort/v5/models/common.py:69-70 (Conv.forward: conv2d -> BatchNorm2d -> SiLU, BN folded),
//           common.py:115-116 (Bottleneck residual), yolort/models/box_head.py:36,74 (head conv).
#include "conv_igemm_impl.hpp"

namespace ymi {

// instantiated in conv_inst_*.hip / head_inst_*.hip
#ifndef YMI_MONOLITHIC
extern template int launch_tile_group<0, YMI_F16, YMI_F16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<1, YMI_F16, YMI_F16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<2, YMI_F16, YMI_F16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<3, YMI_F16, YMI_F16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<4, YMI_F16, YMI_F16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<0, YMI_F16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<1, YMI_F16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<2, YMI_F16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<3, YMI_F16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<4, YMI_F16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<0, YMI_BF16, YMI_BF16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<1, YMI_BF16, YMI_BF16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<2, YMI_BF16, YMI_BF16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<3, YMI_BF16, YMI_BF16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<4, YMI_BF16, YMI_BF16>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<0, YMI_BF16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<1, YMI_BF16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<2, YMI_BF16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<3, YMI_BF16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_tile_group<4, YMI_BF16, YMI_F32>(const ConvArgs&, bool, int, hipStream_t);
extern template int launch_head_decode<YMI_F16, 1>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_F16, 1>(const HeadGroupArgs&, hipStream_t);
extern template int launch_head_decode<YMI_F16, 2>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_F16, 2>(const HeadGroupArgs&, hipStream_t);
extern template int launch_head_decode<YMI_F16, 3>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_F16, 3>(const HeadGroupArgs&, hipStream_t);
extern template int launch_head_decode<YMI_F16, 4>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_F16, 4>(const HeadGroupArgs&, hipStream_t);
extern template int launch_head_decode<YMI_BF16, 1>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_BF16, 1>(const HeadGroupArgs&, hipStream_t);
extern template int launch_head_decode<YMI_BF16, 2>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_BF16, 2>(const HeadGroupArgs&, hipStream_t);
extern template int launch_head_decode<YMI_BF16, 3>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_BF16, 3>(const HeadGroupArgs&, hipStream_t);
extern template int launch_head_decode<YMI_BF16, 4>(const ConvArgs&, const HeadDecodeArgs&, hipStream_t);
extern template int launch_head_group<YMI_BF16, 4>(const HeadGroupArgs&, hipStream_t);
#endif

int conv3x3_rs_launch(const ConvArgs& a, int dtype, int out_dtype, int variant, hipStream_t s);   // conv3x3_rs.hip (declared here: conv_common.hpp is every unit's header)

template <int DT, int ODT>
static int launch_dtype(const ConvArgs& a0, bool is1x1, int tile, hipStream_t s) {
    ConvArgs a = a0;
    a.debug = 0;
    if (tile >= 0x100) { a.debug = tile >> 8; tile &= 0xff; }
    const bool force_v1 = tile < 0;
    if (force_v1) tile = -tile == 100 ? 0 : -tile;   // negative tile ids force the register-staged kernel (-100 = auto)
    if (a.chain_w != nullptr && (force_v1 || (tile >= 1 && tile <= 5 && a.zeros == nullptr) || tile == 41)) {
        set_error("ymi_conv2d: the chained 1x1 convolution needs the pipelined implicit-GEMM kernel");
        return YMI_EINVAL;
    }
    if (tile == 0 && a.chain_w != nullptr) tile = a.chain_k == 32 ? 3 : (a.chain_k == 64 ? 2 : 78);   // cout width == chain K, four waves along the pixels
    if (tile == 0) {
        const int cp = a.cout_pad;
        if (cp <= 32) tile = 3;
        else if (cp <= 64) tile = 2;
        else {
            // enough 128x128 tiles to fill 256 CUs a few times over? else halve the pixel tile
            const long blocks = (long)cdiv(a.M, 128) * cdiv(cp, 128);
            tile = (blocks >= 512) ? 1 : 4;
            if (cp % 128 != 0 && cp % 64 == 0 && cp < 128) tile = 2;
        }
    }
    if (tile < 10 && a.zeros != nullptr && !force_v1) tile += 10;   // pipelined kernel whenever a zero page is supplied
    if (tile >= 10 && a.zeros == nullptr) {
        set_error("ymi_conv2d: tile %d (pipelined kernel) needs desc.zeros", tile);
        return YMI_EINVAL;
    }
    if (tile >= 31 && tile <= 39) return conv3x3_halo_launch(a, DT, ODT, tile - 30, s);   // LDS-halo 3x3 s1 kernel
    if (tile == 41) return conv_stem_launch(a, DT, ODT, s);
    if (tile >= 91 && tile <= 99) return conv_halo8_launch(a, DT, ODT, tile - 90, s);     // 8-wave LDS-halo 3x3 s1 kernel
    if (tile >= 111 && tile <= 120) return conv_igemm8_launch(a, DT, ODT, tile - 110, s);  // 8-wave implicit GEMM, 64-deep steps (120: 192-cout blocks)
    if (tile >= 151 && tile <= 159) return conv_igemm8_launch(a, DT, ODT, tile - 140, s);  // ... with row-transposed stores (variants 11 .. 19)
    if (tile >= 121 && tile <= 124) return conv1x1_stream_launch(a, DT, ODT, tile - 120, s);   // streaming 1x1 (cin <= 128), no LDS
    if (tile == 131) return conv3x3_c32_launch(a, DT, ODT, 1, s);                              // resident-weights persistent 3x3, cin = 32
    if (tile == 132) return conv3x3_res_launch(a, DT, ODT, 1, s);                              // ... cin = 48 / 64, stride 1
    if (tile == 133) return conv3x3_rw_launch(a, DT, ODT, 1, s);                               // ... weights in registers (opt-in)
    if (tile == 135) return conv3x3_rw2_launch(a, DT, ODT, 2, s);                              // ... stride 2, cin = 128 -> cout = 128 / 256, K split over two waves (round 4)
    if (tile == 137 || tile == 138) return conv3x3_rs_launch(a, DT, ODT, tile - 136, s);       // row-streaming 3x3, cin = 64: stride 1 -> 64 / stride 2 -> 128 (round 4)
    if (tile == 134) return conv3x3_rw2_launch(a, DT, ODT, 1, s);                              // ... stride 2, cin = 64 -> cout = 128, weights in registers (round 4)
    switch (tile_group_of(tile)) {
        case 0: return launch_tile_group<0, DT, ODT>(a, is1x1, tile, s);
        case 1: return launch_tile_group<1, DT, ODT>(a, is1x1, tile, s);
        case 2: return launch_tile_group<2, DT, ODT>(a, is1x1, tile, s);
        case 4: return launch_tile_group<4, DT, ODT>(a, is1x1, tile, s);
        default: return launch_tile_group<3, DT, ODT>(a, is1x1, tile, s);
    }
}

// descriptor -> kernel arguments (shared by ymi_conv2d and ymi_conv_head_decode); validation of the zero page included
static int fill_conv_args(const ymi_conv_desc* d, ConvArgs& a) {
    a.x = (const uint16_t*)d->x; a.w = (const uint16_t*)d->w; a.bias = d->bias; a.ktab = (const int2*)d->ktab;
    a.y = d->y; a.res = (const uint16_t*)d->res;
    a.n = d->n; a.h = d->h; a.w_in = d->w_in; a.cin = d->cin; a.x_cs = d->x_cstride;
    a.ho = d->ho; a.wo = d->wo; a.cout = d->cout; a.cout_pad = d->cout_pad; a.y_cs = d->y_cstride; a.res_cs = d->res_cstride;
    a.sh = d->sh; a.sw = d->sw; a.ph = d->ph; a.pw = d->pw; a.k_pad = d->k_pad; a.act = d->act;
    a.M = d->n * d->ho * d->wo; a.nblk_m = 0; a.nblk_n = 0;
    a.y2 = d->y2; a.y2_cs = d->y2_cstride; a.split = d->cout_split; a.zeros = (const uint16_t*)d->zeros;
    a.up2 = d->y2_mode == 1 ? 1 : 0;
    a.chain_w = (const uint16_t*)d->chain_w; a.chain_bias = d->chain_bias; a.chain_y = d->chain_y;
    a.chain_cout = d->chain_cout; a.chain_y_cs = d->chain_y_cstride; a.chain_k = d->cout_split > 0 ? d->cout_split : d->cout;
    a.chain_x2 = (const uint16_t*)d->chain_x2; a.chain_x2_cs = d->chain_x2_cstride; a.chain_k2 = d->chain_x2 != nullptr ? d->chain_k2 : 0;
    a.kh = d->kh; a.kw = d->kw; a.x_zero_off = 0;
    auto magic = [](int dv) { const uint64_t v = (((uint64_t)1 << 32) / (uint64_t)dv) + 1u; return (unsigned)(v > 0xffffffffull ? 0xffffffffull : v); };
    a.magic_hw = magic(d->ho * d->wo);
    a.magic_w = magic(d->wo);
    if (d->zeros != nullptr) {
        const int64_t dz = ((const char*)d->zeros - (const char*)d->x) / 2;
        YMI_REQUIRE(dz > -((int64_t)1 << 31) && dz < ((int64_t)1 << 31) && ((const char*)d->zeros - (const char*)d->x) % 16 == 0,
                    "ymi_conv2d: desc.zeros must lie within +-4 GiB of x and be 16-byte aligned relative to it (use the tail of x's buffer)");
        a.x_zero_off = (int)dz;
    }
    return YMI_OK;
}

int conv2d_launch(const ymi_conv_desc* d, hipStream_t s) {
    YMI_REQUIRE(d != nullptr, "ymi_conv2d: null descriptor");
    YMI_REQUIRE(d->x && d->w && d->bias && d->y, "ymi_conv2d: null buffer");
    YMI_REQUIRE(d->cin % 8 == 0 && d->x_cstride % 8 == 0, "ymi_conv2d: cin (%d) and x_cstride (%d) must be multiples of 8", d->cin, d->x_cstride);
    YMI_REQUIRE(d->cout_pad % 32 == 0 && d->k_pad % 32 == 0, "ymi_conv2d: cout_pad (%d) / k_pad (%d) must be multiples of 32", d->cout_pad, d->k_pad);
    YMI_REQUIRE(d->k_pad >= d->kh * d->kw * d->cin, "ymi_conv2d: k_pad %d < K %d", d->k_pad, d->kh * d->kw * d->cin);
    YMI_REQUIRE(d->y_cstride % 4 == 0 && (d->res == nullptr || d->res_cstride % 4 == 0), "ymi_conv2d: y/res cstride must be multiples of 4");
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16 || d->dtype == YMI_F32, "ymi_conv2d: dtype must be F16, BF16 or F32 (parity mode)");
    YMI_REQUIRE(d->act == YMI_ACT_NONE || d->act == YMI_ACT_SILU, "ymi_conv2d: the convolution epilogues carry SiLU / identity only (activation %d: run the convolution with YMI_ACT_NONE and ymi_act over its output)", d->act);
    YMI_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == YMI_F32, "ymi_conv2d: out_dtype must equal dtype or be F32");
    YMI_REQUIRE(d->ho == (d->h + 2 * d->ph - d->kh) / d->sh + 1 && d->wo == (d->w_in + 2 * d->pw - d->kw) / d->sw + 1,
                "ymi_conv2d: output size %dx%d inconsistent with input %dx%d k%dx%d s%dx%d p%dx%d", d->ho, d->wo, d->h, d->w_in, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
    const bool is1x1 = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0;
    YMI_REQUIRE(is1x1 || d->ktab != nullptr, "ymi_conv2d: ktab required for non-1x1 convolutions");
    if ((int64_t)d->n * d->ho * d->wo >= (int64_t)1 << 31) {
        set_error("ymi_conv2d: too many output pixels");
        return YMI_EINVAL;
    }
    ConvArgs a;
    { const int rc_args = fill_conv_args(d, a); if (rc_args != YMI_OK) return rc_args; }
    YMI_REQUIRE(a.split == 0 || (d->y2 != nullptr && a.split % 8 == 0 && a.split < d->cout && d->res == nullptr && d->out_dtype == d->dtype && d->y2_cstride % 8 == 0),
                "ymi_conv2d: invalid second-output configuration");
    YMI_REQUIRE(a.split == 0 || a.zeros != nullptr || d->dtype == YMI_F32, "ymi_conv2d: the second output needs the pipelined kernel (desc.zeros)");
    YMI_REQUIRE(d->y2_mode == 0 || d->y2_mode == 1, "ymi_conv2d: unknown y2_mode %d", d->y2_mode);
    if (d->chain_w != nullptr) {
        const int k1 = d->cout_split > 0 ? d->cout_split : d->cout;
        YMI_REQUIRE(d->chain_bias && d->chain_y && (k1 == 32 || k1 == 64 || k1 == 128) && d->chain_cout % 32 == 0 && d->chain_cout >= 32 && d->chain_cout <= 128 &&
                        d->chain_y_cstride % 8 == 0 && d->act == YMI_ACT_SILU && d->out_dtype == d->dtype && d->y2_mode == 0 && a.zeros != nullptr,
                    "ymi_conv2d: chained 1x1 needs chain_bias / chain_y, K1 in {32, 64, 128}, chain_cout %% 32 == 0 (<= 128), SiLU, a 16-bit output, desc.zeros");
        YMI_REQUIRE(d->chain_x2 == nullptr || (d->chain_k2 % 16 == 0 && d->chain_k2 >= 16 && d->chain_k2 <= 128 && d->chain_x2_cstride % 8 == 0 && d->cout_split == 0),
                    "ymi_conv2d: chained conv second source: chain_k2 %% 16 == 0 (16..128), chain_x2_cstride %% 8 == 0, no channel split");
        YMI_REQUIRE(d->res == nullptr || d->chain_x2 != nullptr || true, "ymi_conv2d: internal");
    }
    YMI_REQUIRE(d->y2_mode == 0 || (d->y2 != nullptr && a.split == 0 && d->cout % 32 == 0 && d->out_dtype == d->dtype && d->y2_cstride % 8 == 0 && a.zeros != nullptr),
                "ymi_conv2d: the upsampled second output needs y2, cout_split == 0, cout %% 32 == 0, a 16-bit output, y2_cstride %% 8 == 0 and desc.zeros");
    if (d->y2_mode == 1) {
        YMI_REQUIRE(d->tile >= 0, "ymi_conv2d: the upsampled second output is not available in the register-staged kernel");
        YMI_REQUIRE((int64_t)d->n * 4 * d->ho * d->wo < ((int64_t)1 << 31), "ymi_conv2d: upsampled view too large");
    }
    if (a.zeros != nullptr) {
        // 32-bit element offsets inside the pipelined kernel
        YMI_REQUIRE((int64_t)d->n * d->h * d->w_in * d->x_cstride < ((int64_t)1 << 31) && (int64_t)d->cout_pad * d->k_pad < ((int64_t)1 << 31),
                    "ymi_conv2d: tensor too large for the pipelined kernel's 32-bit offsets");
        YMI_REQUIRE(d->y_cstride % 8 == 0 || d->out_dtype == YMI_F32, "ymi_conv2d: y_cstride must be a multiple of 8");
    }
    if (a.M == 0) return YMI_OK;
    if (d->dtype == YMI_F32) {   // fp32 mode: exact fp32 arithmetic.  LDS-DMA pipelined tiles (conv_f32_pipe.hip) whenever the zero page is supplied;
        // a negative tile id (or no zero page: foreign input buffers) selects the register-staged kernel of rounds 2-4 (conv_f32.hip)
        YMI_REQUIRE(d->chain_w == nullptr, "ymi_conv2d: the fp32 mode has no chained convolution");
        if (d->tile >= 0 && a.zeros != nullptr) return conv_f32_pipe_launch(a, is1x1, d->tile, s);
        YMI_REQUIRE(d->y2_mode == 0, "ymi_conv2d: the register-staged fp32 kernel has no upsampled second output");
        return conv_f32_launch(a, is1x1, s);
    }
    if (d->dtype == YMI_F16) {
        if (d->out_dtype == YMI_F32) return launch_dtype<YMI_F16, YMI_F32>(a, is1x1, d->tile, s);
        return launch_dtype<YMI_F16, YMI_F16>(a, is1x1, d->tile, s);
    } else {
        if (d->out_dtype == YMI_F32) return launch_dtype<YMI_BF16, YMI_F32>(a, is1x1, d->tile, s);
        return launch_dtype<YMI_BF16, YMI_BF16>(a, is1x1, d->tile, s);
    }
}


// validation + argument construction of one level's fused head (shared by the single and the grouped launch)
static int head_decode_prepare(const ymi_conv_desc* d, const ymi_post_desc* post, int level, ConvArgs& a, HeadDecodeArgs& h, int& tna) {
    YMI_REQUIRE(d != nullptr && post != nullptr, "ymi_conv_head_decode: null descriptor");
    YMI_REQUIRE(level >= 0 && level < post->num_levels && post->num_levels <= YMI_MAX_LEVELS, "ymi_conv_head_decode: level %d out of range", level);
    YMI_REQUIRE(d->x && d->w && d->bias && d->zeros, "ymi_conv_head_decode: null buffer (x, w, bias and the zero page are required)");
    const int K = post->num_classes + 5;
    const int ra = (K + 31) / 32 * 32;
    YMI_REQUIRE(ra <= 128, "ymi_conv_head_decode: %d outputs per anchor exceed 128 (use ymi_conv2d + ymi_postprocess)", K);
    YMI_REQUIRE(d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && d->cin % 32 == 0 && d->k_pad == d->cin,
                "ymi_conv_head_decode: a 1x1 stride-1 convolution with cin %% 32 == 0 and k_pad == cin is required");
    YMI_REQUIRE(d->cout == 3 * ra && d->cout_pad == 3 * ra, "ymi_conv_head_decode: cout / cout_pad must be 3 x %d (anchor-padded packing)", ra);
    YMI_REQUIRE(d->res == nullptr && d->cout_split == 0 && d->act == YMI_ACT_NONE && d->chain_w == nullptr && d->y2_mode == 0,
                "ymi_conv_head_decode: no residual / second output / chained conv / activation");
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16, "ymi_conv_head_decode: dtype must be F16 or BF16");
    YMI_REQUIRE(d->x_cstride % 8 == 0, "ymi_conv_head_decode: x_cstride must be a multiple of 8");
    YMI_REQUIRE(d->n == post->n && d->ho == post->lh[level] && d->wo == post->lw[level] && d->h == d->ho && d->w_in == d->wo,
                "ymi_conv_head_decode: conv geometry %dx%dx%d does not match level %d of the post-process (%dx%dx%d)", d->n, d->ho, d->wo, level, post->n,
                post->lh[level], post->lw[level]);
    YMI_REQUIRE((int64_t)d->n * d->h * d->w_in * d->x_cstride < ((int64_t)1 << 31) && (int64_t)d->n * d->ho * d->wo < ((int64_t)1 << 31),
                "ymi_conv_head_decode: tensor too large for 32-bit offsets");
    YMI_REQUIRE(post->status && post->ws && post->n >= 1 && post->cand_cap >= 1, "ymi_conv_head_decode: incomplete post-process descriptor");
    const PostLayout L = post_layout(post);
    YMI_REQUIRE(L.label_bits + L.anchor_bits <= 32, "ymi_conv_head_decode: candidate index exceeds 32 bits");
    const Workspace w = carve(post->ws, post->n, L.total_anchors, post->cand_cap);
    YMI_REQUIRE(post->ws_bytes >= w.total, "ymi_conv_head_decode: workspace too small");
    { const int rc_args = fill_conv_args(d, a); if (rc_args != YMI_OK) return rc_args; }
    a.nblk_m = cdiv(a.M, 128);
    a.nblk_n = head_anchor_split() ? 3 : 1;
    { static const int dbg = [] { const char* e = getenv("YOLORT_AMD_HEAD_DEBUG"); return e ? atoi(e) : 0; }(); a.debug = dbg; }   // tuning aid (results are garbage when set)
    h.stride = post->stride[level];
    for (int k = 0; k < 6; ++k) h.anc[k] = post->anchors[level][k];
    h.K = K;
    h.level_off = 0;
    for (int l = 0; l < level; ++l) h.level_off += 3 * post->lh[l] * post->lw[l];
    h.sink = make_sink(post, w, L);
    tna = ra / 32;
    return YMI_OK;
}

#define YMI_HD_DISPATCH(FN, DTYPE, TNA_, ...)                                                              \
    switch (TNA_) {                                                                                        \
        case 1: return DTYPE == YMI_F16 ? FN<YMI_F16, 1>(__VA_ARGS__) : FN<YMI_BF16, 1>(__VA_ARGS__);     \
        case 2: return DTYPE == YMI_F16 ? FN<YMI_F16, 2>(__VA_ARGS__) : FN<YMI_BF16, 2>(__VA_ARGS__);     \
        case 3: return DTYPE == YMI_F16 ? FN<YMI_F16, 3>(__VA_ARGS__) : FN<YMI_BF16, 3>(__VA_ARGS__);     \
        case 4: return DTYPE == YMI_F16 ? FN<YMI_F16, 4>(__VA_ARGS__) : FN<YMI_BF16, 4>(__VA_ARGS__);     \
    }

// Head conv of pyramid level `level` with decode + threshold fused into the epilogue.  The descriptor's weights hold
// every anchor's K rows padded to RA = round_up(K, 32) rows (cout = cout_pad = 3*RA); y is not written.
int conv_head_decode_launch(const ymi_conv_desc* d, const ymi_post_desc* post, int level, hipStream_t s) {
    ConvArgs a;
    HeadDecodeArgs h;
    int tna = 0;
    const int rc = head_decode_prepare(d, post, level, a, h, tna);
    if (rc != YMI_OK) return rc;
    if (a.M == 0) return YMI_OK;
    YMI_HD_DISPATCH(launch_head_decode, d->dtype, tna, a, h, s)
    set_error("ymi_conv_head_decode: unsupported anchor padding");
    return YMI_EINVAL;
}

// all levels in one launch (descs[l] belongs to level l of `post`)
int conv_head_decode_group_launch(const ymi_conv_desc* descs, int n_levels, const ymi_post_desc* post, hipStream_t s) {
    YMI_REQUIRE(descs != nullptr && post != nullptr && n_levels >= 1 && n_levels <= YMI_MAX_LEVELS && n_levels == post->num_levels,
                "ymi_conv_head_decode_group: one descriptor per pyramid level of the post-process is required");
    HeadGroupArgs g;
    memset(&g, 0, sizeof(g));
    g.n = n_levels;
    int tna0 = 0, blocks = 0;
    for (int l = 0; l < n_levels; ++l) {
        int tna = 0;
        const int rc = head_decode_prepare(&descs[l], post, l, g.a[l], g.h[l], tna);
        if (rc != YMI_OK) return rc;
        YMI_REQUIRE(descs[l].dtype == descs[0].dtype && (l == 0 || tna == tna0), "ymi_conv_head_decode_group: levels must share dtype and class count");
        tna0 = tna;
    }
    // coarse levels first: they have the longest K loops and the fewest blocks, so they should not form the tail
    for (int l = n_levels - 1; l >= 0; --l) {
        g.first_block[l] = blocks;
        blocks += g.a[l].nblk_m * g.a[l].nblk_n;
    }
    g.first_block[n_levels] = blocks;
    if (blocks == 0) return YMI_OK;
    YMI_HD_DISPATCH(launch_head_group, descs[0].dtype, tna0, g, s)
    set_error("ymi_conv_head_decode_group: unsupported anchor padding");
    return YMI_EINVAL;
}
#undef YMI_HD_DISPATCH

}  // namespace ymi

#ifdef YMI_STAMPS
extern "C" int ymi_debug_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ymi::ymi_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ymi_conv2d(const ymi_conv_desc* d, void* stream) { return ymi::conv2d_launch(d, (hipStream_t)stream); }
extern "C" int ymi_conv_stem_planar(const ymi_conv_desc* d, const void* const* imgs, int n_imgs, void* stream) {
    using namespace ymi;
    YMI_REQUIRE(d != nullptr && imgs != nullptr, "ymi_conv_stem_planar: null argument");
    YMI_REQUIRE(d->w && d->bias && d->y && d->zeros, "ymi_conv_stem_planar: null buffer (w, bias, y and the zero page are required)");
    YMI_REQUIRE(n_imgs == d->n && n_imgs >= 1, "ymi_conv_stem_planar: %d images for a descriptor of batch %d", n_imgs, d->n);
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16, "ymi_conv_stem_planar: dtype must be F16 or BF16");
    YMI_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == YMI_F32, "ymi_conv_stem_planar: out_dtype must equal dtype or be F32");
    YMI_REQUIRE(d->ho == (d->h + 2 * d->ph - d->kh) / d->sh + 1 && d->wo == (d->w_in + 2 * d->pw - d->kw) / d->sw + 1, "ymi_conv_stem_planar: inconsistent output size");
    YMI_REQUIRE(d->y_cstride % 8 == 0 || d->out_dtype == YMI_F32, "ymi_conv_stem_planar: y_cstride must be a multiple of 8");
    for (int i = 0; i < n_imgs; ++i) YMI_REQUIRE(imgs[i] != nullptr && ((uintptr_t)imgs[i] & 15) == 0, "ymi_conv_stem_planar: image %d is null or not 16-byte aligned", i);
    ymi_conv_desc dd = *d;
    dd.x = d->zeros;   // not read; keeps the zero-page offset check of the shared argument builder trivially true
    ConvArgs a;
    { const int rc_args = fill_conv_args(&dd, a); if (rc_args != YMI_OK) return rc_args; }
    return conv_stem_planar_launch(a, imgs, d->dtype, d->out_dtype, (hipStream_t)stream);
}

extern "C" int ymi_stem_body1_planar(const ymi_conv_desc* stem, const ymi_conv_desc* body1, const void* const* imgs, int n_imgs, void* stream) {
    using namespace ymi;
    YMI_REQUIRE(stem != nullptr && body1 != nullptr && imgs != nullptr, "ymi_stem_body1_planar: null argument");
    YMI_REQUIRE(stem->w && stem->bias && stem->zeros && body1->w && body1->bias && body1->y, "ymi_stem_body1_planar: null buffer (both weights / biases, the zero page and body1.y are required)");
    YMI_REQUIRE(n_imgs == stem->n && n_imgs == body1->n && n_imgs >= 1, "ymi_stem_body1_planar: %d images for descriptors of batch %d / %d", n_imgs, stem->n, body1->n);
    YMI_REQUIRE((stem->dtype == YMI_F16 || stem->dtype == YMI_BF16) && body1->dtype == stem->dtype && stem->out_dtype == stem->dtype && body1->out_dtype == stem->dtype,
                "ymi_stem_body1_planar: both convolutions must compute and store F16 or BF16");
    YMI_REQUIRE(stem->ho == (stem->h + 2 * stem->ph - stem->kh) / stem->sh + 1 && stem->wo == (stem->w_in + 2 * stem->pw - stem->kw) / stem->sw + 1, "ymi_stem_body1_planar: inconsistent stem output size");
    YMI_REQUIRE(body1->y_cstride % 8 == 0, "ymi_stem_body1_planar: y_cstride must be a multiple of 8");
    for (int i = 0; i < n_imgs; ++i) YMI_REQUIRE(imgs[i] != nullptr && ((uintptr_t)imgs[i] & 15) == 0, "ymi_stem_body1_planar: image %d is null or not 16-byte aligned", i);
    ymi_conv_desc d1 = *stem, d2 = *body1;
    d1.x = stem->zeros;                              // neither input pointer is read: keep the zero-page offset check of the shared argument builder trivially true
    d1.y = body1->y;
    d2.x = body1->zeros ? body1->zeros : stem->zeros;
    d2.zeros = d2.x;
    ConvArgs a1, a2;
    { const int rc_args = fill_conv_args(&d1, a1); if (rc_args != YMI_OK) return rc_args; }
    { const int rc_args = fill_conv_args(&d2, a2); if (rc_args != YMI_OK) return rc_args; }
    return stem_body1_planar_launch(a1, a2, imgs, stem->dtype, (hipStream_t)stream);
}

namespace ymi {
// ops 0 + 1 of a plan (the stem reading the letterboxed canvas, body.1 reading the stem's output) as one launch; the stem's y is not written
int stem_body1_desc_launch(const ymi_conv_desc* stem, const ymi_conv_desc* body1, hipStream_t s) {
    YMI_REQUIRE(stem != nullptr && body1 != nullptr, "ymi_stem_body1: null argument");
    YMI_REQUIRE(stem->x && stem->w && stem->bias && stem->zeros && body1->w && body1->bias && body1->y, "ymi_stem_body1: null buffer (stem.x, both weights / biases, the zero page and body1.y are required)");
    YMI_REQUIRE(stem->n == body1->n && stem->n >= 1, "ymi_stem_body1: descriptors of batch %d / %d", stem->n, body1->n);
    YMI_REQUIRE((stem->dtype == YMI_F16 || stem->dtype == YMI_BF16) && body1->dtype == stem->dtype && stem->out_dtype == stem->dtype && body1->out_dtype == stem->dtype,
                "ymi_stem_body1: both convolutions must compute and store F16 or BF16");
    YMI_REQUIRE(stem->ho == (stem->h + 2 * stem->ph - stem->kh) / stem->sh + 1 && stem->wo == (stem->w_in + 2 * stem->pw - stem->kw) / stem->sw + 1, "ymi_stem_body1: inconsistent stem output size");
    YMI_REQUIRE(body1->y_cstride % 8 == 0, "ymi_stem_body1: y_cstride must be a multiple of 8");
    ymi_conv_desc d1 = *stem, d2 = *body1;
    d1.y = body1->y;                                 // the stem's output is not written
    d2.x = body1->zeros ? body1->zeros : stem->zeros;   // not read: keep the zero-page offset check of the shared argument builder trivially true
    d2.zeros = d2.x;
    ConvArgs a1, a2;
    { const int rc_args = fill_conv_args(&d1, a1); if (rc_args != YMI_OK) return rc_args; }
    { const int rc_args = fill_conv_args(&d2, a2); if (rc_args != YMI_OK) return rc_args; }
    return stem_body1_launch(a1, a2, stem->dtype, s);
}
}  // namespace ymi

extern "C" int ymi_stem_body1(const ymi_conv_desc* stem, const ymi_conv_desc* body1, void* stream) {
    return ymi::stem_body1_desc_launch(stem, body1, (hipStream_t)stream);
}

extern "C" int ymi_conv_head_decode_group(const ymi_conv_desc* convs, int n_levels, const ymi_post_desc* post, void* stream) {
    return ymi::conv_head_decode_group_launch(convs, n_levels, post, (hipStream_t)stream);
}

extern "C" int ymi_conv_head_decode(const ymi_conv_desc* conv, const ymi_post_desc* post, int level, void* stream) {
    return ymi::conv_head_decode_launch(conv, post, level, (hipStream_t)stream);
}

extern "C" int ymi_conv_f32_pick_tile(int m_pixels, int cout_pad) { return ymi::conv_f32_pick_tile(m_pixels, cout_pad); }
extern "C" int ymi_conv_build_ktab(int cin, int kh, int kw, int w_in, int x_cstride, int k_pad, int32_t* t) {
    if (cin % 8 != 0 || k_pad % 32 != 0 || t == nullptr) {
        ymi::set_error("ymi_conv_build_ktab: cin %% 8 and k_pad %% 32 must be 0");
        return YMI_EINVAL;
    }
    const int K = kh * kw * cin;
    for (int q = 0; q < k_pad / 8; ++q) {
        const int k0 = q * 8;
        if (k0 >= K) {
            t[2 * q] = 0;
            t[2 * q + 1] = -1;
            continue;
        }
        const int tap = k0 / cin, c = k0 - tap * cin;
        const int dy = tap / kw, dx = tap - dy * kw;
        t[2 * q] = (dy * w_in + dx) * x_cstride + c;
        t[2 * q + 1] = (dy << 16) | dx;
    }
    return YMI_OK;
}
