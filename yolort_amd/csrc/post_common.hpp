// Shared pieces of the detection post-process (postprocess.hip) and of the head convolution with the
// decode fused into its epilogue (conv_igemm.hip / head_decode.hpp): workspace layout, status words,
// candidate sink and the box-decode arithmetic.
#pragma once
#include "common.hpp"

namespace ymi {

// status words (device int32[4])
enum { ST_NCAND = 0, ST_OVERFLOW = 1, ST_NSEG = 2, ST_RSV = 3, ST_RAW = 4, ST_DONE = 5, ST_WORDS = 8 };

struct Workspace {
    // all device pointers, carved from the caller's `ws`
    float* boxes_all;     // (n, A, 4) decoded xyxy boxes of every anchor
    uint64_t* hi[2];      // ping-pong record arrays (cand_cap each)
    uint32_t* lo[2];
    uint32_t* hist;       // (max_blocks, 256) per-block digit counts / offsets
    uint32_t* seg_start;  // (cand_cap) segment start positions (unordered)
    float* kept_box;      // (cand_cap, 4) per-segment kept boxes
    uint8_t* keep;        // (cand_cap) keep flag indexed by global rank r
    int* img_count;       // (n) candidates appended per image (per-image sort path)
    uint64_t* p_hi;       // per-class order P of the per-image sort path (cand_cap each)
    uint32_t* p_lo;
    int* sel_count;       // (2n) per-image sort path: records kept by the score-prefix selection, then [n..2n) truncated flags
    int64_t total;
};

constexpr int SORT_ITEMS = 2048;  // records per block per radix pass (256 threads x 8)

inline int64_t align_up(int64_t v) { return (v + 255) & ~(int64_t)255; }

inline Workspace carve(void* ws, int n, int total_anchors, int cand_cap) {
    Workspace w;
    char* p = (char*)ws;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { char* q = p ? p + off : nullptr; off += align_up(bytes); return q; };
    const int max_blocks = cdiv(cand_cap, SORT_ITEMS);
    w.boxes_all = (float*)take((int64_t)n * total_anchors * 16);
    w.hi[0] = (uint64_t*)take((int64_t)cand_cap * 8);
    w.hi[1] = (uint64_t*)take((int64_t)cand_cap * 8);
    w.lo[0] = (uint32_t*)take((int64_t)cand_cap * 4);
    w.lo[1] = (uint32_t*)take((int64_t)cand_cap * 4);
    w.hist = (uint32_t*)take((int64_t)max_blocks * 256 * 4 + 1024);
    w.seg_start = (uint32_t*)take((int64_t)cand_cap * 4);
    w.kept_box = (float*)take((int64_t)cand_cap * 16);
    w.keep = (uint8_t*)take((int64_t)cand_cap);
    w.img_count = (int*)take((int64_t)(n > 0 ? n : 1) * 4);
    w.p_hi = (uint64_t*)take((int64_t)cand_cap * 8);
    w.p_lo = (uint32_t*)take((int64_t)cand_cap * 4);
    w.sel_count = (int*)take((int64_t)(n > 0 ? n : 1) * 8);
    w.total = off;
    return w;
}

inline int bits_for(int64_t v) {  // number of bits needed to represent values in [0, v)
    int b = 0;
    while (((int64_t)1 << b) < v) ++b;
    return b;
}

// where candidate records (hi = img << 32 | ~score_bits, lo = anchor << label_bits | label) and decoded boxes go
struct CandSink {
    float* boxes_all;      // (n, total_anchors, 4) xyxy of EVERY anchor (NMS looks boxes up by anchor index)
    uint64_t* hi;
    uint32_t* lo;
    int* status;
    int cap;               // global record capacity (global-sort path)
    int* img_count;        // non-null: append to per-image regions [img*cap_img, +cap_img) (per-image LDS sort path)
    int cap_img;
    int total_anchors;     // anchors per image over all levels
    int label_bits;
    float thr;             // score threshold (strict >)
};

// layout decisions shared by every producer / consumer of the records of one ymi_post_desc
struct PostLayout {
    int total_anchors, label_bits, anchor_bits, cap_img;
    bool per_image;
};
inline PostLayout post_layout(const ymi_post_desc* d) {
    PostLayout L;
    L.total_anchors = 0;
    for (int l = 0; l < d->num_levels; ++l) L.total_anchors += 3 * d->lh[l] * d->lw[l];
    L.label_bits = bits_for(d->num_classes) < 1 ? 1 : bits_for(d->num_classes);
    L.anchor_bits = bits_for(L.total_anchors);
    // per-image LDS sort when an image's capacity is large enough; every image owns a power-of-two sized region
    L.cap_img = 64;
    while (L.cap_img * 2 <= d->cand_cap / d->n) L.cap_img *= 2;
    L.per_image = d->cand_cap / d->n >= 64;
    return L;
}
inline CandSink make_sink(const ymi_post_desc* d, const Workspace& w, const PostLayout& L) {
    CandSink k;
    k.boxes_all = w.boxes_all; k.hi = w.hi[0]; k.lo = w.lo[0]; k.status = d->status; k.cap = d->cand_cap;
    k.img_count = L.per_image ? w.img_count : nullptr; k.cap_img = L.cap_img;
    k.total_anchors = L.total_anchors; k.label_bits = L.label_bits; k.thr = d->score_thresh;
    return k;
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// _utils.py:59-60: xy = (s*2 - 0.5 + grid) * stride ; wh = (s*2)**2 * anchor (each op rounded like torch's),
// then box_convert cxcywh -> xyxy (box_head.py:358).  Inputs are the four raw box logits.
__device__ __forceinline__ f32x4 decode_box(float lx, float ly, float lw, float lh, int x, int y, float stride, float aw, float ah) {
    const float sx = sigmoid_acc(lx), sy = sigmoid_acc(ly), sw = sigmoid_acc(lw), sh = sigmoid_acc(lh);
    const float cx = __fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(sx, 2.0f), 0.5f), (float)x), stride);
    const float cy = __fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(sy, 2.0f), 0.5f), (float)y), stride);
    const float w2 = __fmul_rn(sw, 2.0f), h2 = __fmul_rn(sh, 2.0f);
    const float bw = __fmul_rn(__fmul_rn(w2, w2), aw);
    const float bh = __fmul_rn(__fmul_rn(h2, h2), ah);
    const float hw_ = __fmul_rn(0.5f, bw), hh_ = __fmul_rn(0.5f, bh);
    f32x4 b = {__fsub_rn(cx, hw_), __fsub_rn(cy, hh_), __fadd_rn(cx, hw_), __fadd_rn(cy, hh_)};
    return b;
}

// flush of a wave's LDS record buffer: ONE atomic on the image's (or the global) counter, then a coalesced copy.
// Records past the capacity are dropped but still counted (the host sees the needed capacity in status / img_count).
__device__ __forceinline__ void flush_records(const CandSink& k, const uint64_t* bhi, const uint32_t* blo, int fill, int img, int lane) {
    if (fill == 0) return;
    int base = 0;
    if (k.img_count != nullptr) {
        if (lane == 0) base = atomicAdd(&k.img_count[img], fill);
        base = __shfl(base, 0, 64);
        for (int i = lane; i < fill; i += 64) {
            const int pos = base + i;
            if (pos < k.cap_img) {
                k.hi[(int64_t)img * k.cap_img + pos] = bhi[i];
                k.lo[(int64_t)img * k.cap_img + pos] = blo[i];
            }
        }
    } else {
        if (lane == 0) base = atomicAdd(&k.status[ST_NCAND], fill);
        base = __shfl(base, 0, 64);
        for (int i = lane; i < fill; i += 64) {
            const int pos = base + i;
            if (pos < k.cap) {
                k.hi[pos] = bhi[i];
                k.lo[pos] = blo[i];
            }
        }
    }
}

}  // namespace ymi
