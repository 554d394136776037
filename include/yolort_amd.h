/*
 * yolort_amd.h -- C ABI of libyolort_amd.so, the MI355X (gfx950) YOLOv5 inference hot path.
 *
 * This is the drop-in boundary.  The reference (zhiqwang/yolort) has no C
 * plugin ABI of its own: its plug points are Python constructor injection and ATen/torchvision
 * operators.  Each entry point below therefore replaces one ATen/torchvision call site of the
 * reference's hot path and cites it (paths relative to the reference root).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + explicit sizes, no torch types; `stream` is a hipStream_t
 *     passed as void* (NULL = default stream).  The caller owns every buffer.
 *   - every function returns 0 on success or a negative YMI_E* code and never throws; the message
 *     of the last failure on the calling thread is available from ymi_last_error().
 *   - no hidden hipDeviceSynchronize / hipMalloc; thread-safe for distinct streams.
 *   - activations are NHWC ("pixel-major"): element (n,y,x,c) of a view lives at
 *       base + ((n*H + y)*W + x) * cstride + c        (cstride >= C lets a view be a channel slice
 *     of a wider concat buffer -- this is how torch.cat is eliminated).
 */
#ifndef YOLORT_AMD_H
#define YOLORT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YMI_ABI_VERSION 6

/* error codes */
#define YMI_OK 0
#define YMI_EINVAL -1   /* bad argument / unsupported shape */
#define YMI_EHIP -2     /* HIP runtime error (message has hipGetErrorString) */
#define YMI_ENOMEM -3   /* caller-provided capacity too small (retry with the reported size) */

/* element types */
#define YMI_F16 0
#define YMI_BF16 1
#define YMI_F32 2
#define YMI_U8 3
#define YMI_U8_HWC 4 /* ymi_letterbox input only: interleaved (h, w, 3) uint8, as image decoders deliver it */

/* activation after conv */
#define YMI_ACT_NONE 0
#define YMI_ACT_SILU 1
/* the activations of the legacy r3.1 blocks (common.py:64-65 `Conv(version="r3.1")`: Hardswish; common.py:140 `BottleneckCSP.act`: LeakyReLU(0.1)): NOT carried by the
 * convolution epilogues (ymi_conv2d refuses them) -- such a convolution runs with YMI_ACT_NONE and ymi_act rewrites its output in place */
#define YMI_ACT_HARDSWISH 2
#define YMI_ACT_LEAKY 3

int ymi_abi_version(void);
const char* ymi_last_error(void);
/* number of HIP devices visible, or a negative error; used by the loud "no GPU" check */
int ymi_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Letterbox: per-image aspect-preserving bilinear resize + centred constant pad + dtype cast +
 * CHW -> NHWC(c_out) layout change, one launch for the whole batch.
 * Replaces yolort/models/transform.py:53-97 (_resize_image_and_masks -> F.interpolate bilinear,
 * align_corners=False, recompute_scale_factor=True) and :297-330 (batch_images: new_full + copy_).
 * The host computes resized sizes / pad offsets with the reference's float recipe
 * (yolort_amd.models.transform) and passes them in `geom`.
 *   imgs[i]     device pointer to image i, planar CHW (3,h,w), contiguous, dtype in_dtype
 *               (YMI_F32 / YMI_F16 / YMI_BF16 in [0,1], or YMI_U8 in [0,255] which is scaled by 1/255);
 *               YMI_U8_HWC: interleaved (h,w,3) uint8 instead (the /255, the HWC -> CHW permute of
 *               yolov5.py:218-228 and the letterbox are then one kernel)
 *   geom[i*6..] {h_in, w_in, h_resized, w_resized, pad_top, pad_left}
 *   out         (n, hb, wb, c_out) NHWC, dtype out_dtype, channels 3..c_out-1 zero-filled
 *   fill        pad value (reference: 114/255, transform.py:141)
 * ---------------------------------------------------------------------------------------- */
int ymi_letterbox(const void* const* imgs, const int32_t* geom, int n, int in_dtype, void* out, int hb,
                  int wb, int c_out, int out_dtype, float fill, void* stream);

/* NCHW (n,c,h,w) -> NHWC view and back (module-level API edges; not on the fused path). */
int ymi_nchw_to_nhwc(const void* x, int n, int c, int h, int w, int in_dtype, void* y, int y_cstride,
                     int c_pad, int out_dtype, void* stream);
int ymi_nhwc_to_nchw(const void* x, int x_cstride, int n, int c, int h, int w, int in_dtype, void* y,
                     int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused convolution: y = act(conv2d(x, W) + bias) (+ residual), implicit GEMM on MFMA.
 * Replaces yolort/v5/models/common.py:69-70 `Conv.forward` = SiLU(BN(conv2d(x))) with BatchNorm
 * folded into W/bias on the host (formula of yolort/v5/utils/torch_utils.py:238-245, eps 1e-3),
 * the residual add of common.py:115-116 (Bottleneck) and, with act=NONE and a real bias, the
 * biased 1x1 head conv of yolort/models/box_head.py:36,74.
 *   x        NHWC view (n,h,w,cin) with pixel stride x_cstride; cin % 8 == 0
 *   w        packed weights [cout_pad][k_pad], k index = (ky*kw + kx)*cin + c, dtype = `dtype`,
 *            zero padded; cout_pad % 32 == 0, k_pad % 32 == 0  (see ymi_conv_pack_dims)
 *   bias     fp32 [cout_pad]
 *   ktab     int32 [k_pad/8][2] im2col table: {element offset (dy*w + dx)*x_cstride + c, dy<<16|dx}
 *            for every 8-channel chunk (built by ymi_conv_build_ktab); entries past K are {0,-1}
 *   y        NHWC view (n,ho,wo,cout) with pixel stride y_cstride, dtype out_dtype (dtype or F32)
 *   res      optional residual view, same shape as y, dtype `dtype` (NULL = none); added AFTER act
 *            (not combinable with y2)
 * ---------------------------------------------------------------------------------------- */
typedef struct ymi_conv_desc {
    const void* x;
    const void* w;
    const float* bias;
    const int32_t* ktab;
    void* y;
    const void* res;
    int32_t n, h, w_in, cin, x_cstride;
    int32_t ho, wo, cout, cout_pad, y_cstride, res_cstride;
    int32_t kh, kw, sh, sw, ph, pw, k_pad;
    int32_t act, dtype, out_dtype;
    int32_t tile; /* 0 = auto, else forces a tile configuration (tuning / tests) */
    /* optional second output: output channels [cout_split, cout) go to view y2 (channel 0 of y2 =
     * channel cout_split); lets one launch serve two consumers of the same input (C3.cv1 + C3.cv2,
     * reference common.py:172-173).  cout_split == 0 disables; must be a multiple of 8. */
    void* y2;
    int32_t y2_cstride, cout_split;
    /* y2_mode 0: channel split as above.  y2_mode 1 (cout_split must be 0, cout % 32 == 0, output of the compute dtype): y2 is an
     * (n, 2*ho, 2*wo) view that receives ALL output channels nearest-neighbour upsampled x2, in addition to y --
     * the nn.Upsample(scale_factor=2) of path_aggregation_network.py:221-223 folded into its producer's epilogue. */
    int32_t y2_mode, reserved0;
    /* optional CHAINED 1x1 convolution (Bottleneck.cv1 after C3.cv1, reference common.py:115,172): with K1 = cout_split
     * (or cout when there is no split) a second fused conv t = SiLU(chain_w * y[:, 0:K1] + chain_bias) is evaluated in
     * the epilogue from the freshly rounded outputs still in registers -- one launch instead of two and no re-read of y.
     * chain_w: packed [round_up(chain_cout,128)][K1] like w (K1 in {32, 64, 128}), chain_bias fp32 [chain_cout],
     * chain_y an (n, ho, wo) view of chain_cout channels (chain_cout % 32 == 0, <= 128).  Needs act SILU, a 16-bit
     * output and a tile whose cout width equals K1 (auto tile selection takes care of it); NULL disables. */
    const void* chain_w;
    const float* chain_bias;
    void* chain_y;
    int32_t chain_cout, chain_y_cstride;
    /* chained conv over a CONCAT (C3.cv3 after the last Bottleneck, common.py:173): its input channels are the K1 fresh
     * outputs followed by chain_k2 channels read from the (n, ho, wo) view chain_x2 (the other half of the concat buffer);
     * chain_w rows then hold K1 + chain_k2 weights.  chain_k2 % 16 == 0, <= 128; 0 / NULL = fresh outputs only.  With a
     * second source the producing conv may carry a residual and may be a 3x3 (LDS-halo variants with pixel-major waves). */
    const void* chain_x2;
    int32_t chain_x2_cstride, chain_k2;
    /* >= 256 readable zero bytes in device memory within +-4 GiB of x (e.g. the tail of x's own
     * buffer): source of out-of-image / out-of-range activation chunks for the direct-to-LDS loads of
     * the pipelined kernel, which also requires the packed weight ROWS to be zero-padded to a
     * multiple of 128 (NULL selects the register-staged kernel, which needs neither) */
    const void* zeros;
} ymi_conv_desc;

int ymi_conv2d(const ymi_conv_desc* d, void* stream);
/* fills host array ktab[k_pad/8][2] for the geometry in d (x_cstride, w_in, cin, kh, kw) */
int ymi_conv_build_ktab(int cin, int kh, int kw, int w_in, int x_cstride, int k_pad, int32_t* ktab_host);
/* fp32 mode (dtype = out_dtype = YMI_F32: fp32 activations and weights, exact fp32 arithmetic on the f32-input MFMA -- the
 * arithmetic of the reference's CPU path, common.py:69-70 run in torch.float32): the tile ymi_conv2d takes for tile == 0,
 * a function of (n*ho*wo, cout_pad) only, so every process sums in the same order.  Tiles 201-206 are the LDS-DMA
 * pipelined kernels (need desc.zeros); a negative tile id selects the register-staged kernel of rounds 2-4. */
int ymi_conv_f32_pick_tile(int m_pixels, int cout_pad);

/* ------------------------------------------------------------------------------------------
 * A whole C3 block in one launch: y = cv3(cat(m(cv1(x)), cv2(x))), m = Bottleneck(s)
 * x1 + cv2(cv1(x1)) -- yolort/v5/models/common.py:172-173 (C3.forward) with :115-116
 * (Bottleneck.forward) inlined, every Conv being :69-70 with BatchNorm folded into W/bias.
 * Intermediates never reach memory; results are bit-identical to the separate ymi_conv2d launches.
 * Two instances:
 *   (1) c_in = 64, c_hidden = 32, c_out = 64, one shortcut Bottleneck (yolov5s backbone.body.2): resident weights,
 *       csrc/c3_fused32.hip; wblob = NULL, mode = 0.
 *   (2) ABI 6 -- c_hidden = 64 or 128, c_out = 2 * c_hidden, c_in % 32 == 0: the strip kernel of csrc/c3_tile.hip (strips of whole rows where a
 *       row fits the LDS patch -- 80 x 80 / 40 x 40 of yolov5s --, column tiles with a halo column either side on wider maps; ymi_c3_tile_supported says),
 *       weights streamed in MFMA fragment order from `wblob` (ymi_c3_blob_bytes / ymi_c3_pack build it from w12 .. b3).
 *       A C3 with n > 1 Bottlenecks is a chain of launches over the same descriptor fields (`mode`):
 *         0  whole block, one Bottleneck:   x -> y
 *         1  HEAD  cv1 | cv2 + Bottleneck 0:   x -> y1_out (the Bottleneck's output), y2 (cv2(x))
 *         2  MID   one Bottleneck:             y1_in -> y1_out     (wm1 / wm2 = THAT Bottleneck's weights; never in place)
 *         3  TAIL  last Bottleneck + cv3:      y1_in, y2 -> y
 * Anything else returns YMI_EINVAL.  The Python side emits (1) since round 3 and (2) since round 6
 * (YOLORT_AMD_FUSE_C3=0 / YOLORT_AMD_C3_TILE=0 restore the separate launches).
 *   x / y       NHWC views (n,h,w,c_in) / (n,h,w,c_out), 16-bit, pixel strides x_cstride / y_cstride
 *   w12, b12    cv1 and cv2 stacked along cout: packed [>= 2*c_hidden][k12_pad] like ymi_conv_desc.w
 *               (rows 0..c_hidden-1 = cv1), fp32 bias [2*c_hidden]
 *   wm1, bm1    Bottleneck.cv1 (1x1, c_hidden -> c_hidden);  wm2, bm2  Bottleneck.cv2 (3x3 pad 1,
 *               k = (ky*3 + kx)*c_hidden + c);  w3, b3  cv3 (1x1, 2*c_hidden -> c_out)
 *   shortcut    1: the Bottleneck adds its input (common.py:116 `self.add`)
 * ---------------------------------------------------------------------------------------- */
typedef struct ymi_c3_desc {
    const void* x;
    void* y;
    const void* w12;
    const float* b12;
    const void* wm1;
    const float* bm1;
    const void* wm2;
    const float* bm2;
    const void* w3;
    const float* b3;
    int32_t n, h, w, x_cstride, y_cstride, dtype;
    int32_t c_in, c_hidden, c_out, n_bottlenecks, shortcut;
    int32_t k12_pad, km1_pad, km2_pad, k3_pad;
    int32_t mode;            /* ABI 6 (was reserved0): 0 whole block, 1 HEAD, 2 MID, 3 TAIL */
    /* ABI 6 */
    const void* wblob;       /* instance (2): ymi_c3_pack's weight stream for THIS mode (device memory, 16-byte aligned); NULL for instance (1) */
    const void* y1_in;       /* modes 2, 3: (n,h,w,c_hidden) view, the Bottleneck's input */
    void* y1_out;            /* modes 1, 2: (n,h,w,c_hidden) view, the Bottleneck's output */
    void* y2;                /* mode 1: cv2(x) is written here; mode 3: read from here; (n,h,w,c_hidden) view */
    int32_t y1_in_cstride, y1_out_cstride, y2_cstride, reserved1;
} ymi_c3_desc;

int ymi_c3_fused(const ymi_c3_desc* d, void* stream);
/* instance (2): size of / builder for the fragment-ordered weight stream of descriptor d's (c_in, c_hidden, mode); reads w12 .. b3 (device
 * pointers, the ymi_conv_desc.w layout) and writes `blob` on `stream`.  Weights that a mode does not use (w12 in modes 2 / 3, w3 in 1 / 2) may be NULL. */
int64_t ymi_c3_blob_bytes(const ymi_c3_desc* d);
int ymi_c3_pack(const ymi_c3_desc* d, void* blob, void* stream);
/* 1 when a strip geometry exists for (n, h, w, c_hidden) of d, i.e. ymi_c3_fused would take instance (2) for it */
int ymi_c3_tile_supported(const ymi_c3_desc* d);
/* ... and which one: out6 = {rows per strip R, slot origin delta, column tiles per row band, output columns per tile, patch slots, tiles}; returns 0 (out6 untouched) when
 * none exists.  Column tiles == 1: full-width strips (the rows of the patch share one pad slot); > 1: maps wider than the patch holds, a halo column either side of a tile. */
int ymi_c3_tile_geometry(const ymi_c3_desc* d, int* out6);

/* ------------------------------------------------------------------------------------------
 * SPP max-pool pyramid: given x = channels [0,c) of a (n,h,w,4c) concat buffer, writes
 * maxpool5(x), maxpool9(x), maxpool13(x) (stride 1, "same", -inf padding) into channel slices
 * [c,2c) [2c,3c) [3c,4c).  Replaces common.py:183-187 (SPP.forward: cat([x]+[m(x) for m in self.m])).
 * ---------------------------------------------------------------------------------------- */
int ymi_spp_pool(void* buf, int n, int h, int w, int c, int cstride, int dtype, void* stream);

/* nearest x2 upsample of view x (n,h,w,c) into view y (n,2h,2w,c) -- nn.Upsample(scale_factor=2),
 * yolort/models/path_aggregation_network.py:123,131,134 -- written straight into its concat slot. */
int ymi_upsample2x(const void* x, int x_cstride, int n, int h, int w, int c, void* y, int y_cstride,
                   int dtype, void* stream);
/* y <- act(y) (+ res) in place over the view (npix, c): the legacy r3.1 activations YMI_ACT_HARDSWISH (common.py:64-65) / YMI_ACT_LEAKY (LeakyReLU(0.1), common.py:140)
 * after a convolution run with YMI_ACT_NONE; `res`: the Bottleneck shortcut, added after the activation (common.py:115-116).  c and the strides in multiples of one
 * 16-byte packet (8 halves / 4 floats). */
int ymi_act(void* y, int y_cstride, int npix, int c, int dtype, int act, const void* res, int res_cstride, void* stream);
/* strided channel-slice copy (only used where a producer cannot write into its concat slot) */
int ymi_copy_view(const void* x, int x_cstride, int npix, int c, void* y, int y_cstride, int dtype,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-process (yolort/models/box_head.py:328-360,414-427 + torchvision batched_nms).
 * ymi_postprocess runs, on `stream`, with no host sync:
 *   1. decode   sigmoid, xy=(s*2-0.5+grid)*stride, wh=(s*2)^2*anchor (yolort/models/_utils.py:59-60,
 *               grids/shifts of anchor_utils.py in closed form), score = cls*obj (box_head.py:357),
 *               strict threshold score > score_thresh over ALL (anchor,class) pairs (box_head.py:418)
 *   2. sort     candidates per image by score descending, ties by candidate index (anchor asc,
 *               class asc) -- the stable order of torchvision.ops.nms
 *   3. nms      class-aware greedy suppression, IoU > nms_thresh strict, fp32 (box_head.py:422)
 *   4. top-k    first detections_per_img kept (box_head.py:424) and box rescale to the original
 *               image (yolort/models/transform.py:354-367 scale_coords, no clipping)
 * Inputs: logits[l] = fp32 NHWC (n, H_l, W_l, lcstride) with channel a*K + k (K = num_classes+5).
 * Outputs (fixed slab, the reference TensorRT wire format of relay/trt_graphsurgeon.py:223-244):
 *   out_boxes (n, K, 4) fp32, out_scores (n, K) fp32, out_labels (n, K) int64, out_count (n) int32.
 * Workspace: `ws` of `ws_bytes` bytes (query with ymi_postprocess_ws_bytes); candidate capacity
 * `cand_cap` is the batch-wide maximum number of (anchor,class) candidates; on overflow nothing is
 * truncated silently: the needed count is left in status[0] and status[1] is set to 1.
 * ---------------------------------------------------------------------------------------- */
#define YMI_MAX_LEVELS 4
typedef struct ymi_post_desc {
    const float* logits[YMI_MAX_LEVELS];
    int32_t lh[YMI_MAX_LEVELS], lw[YMI_MAX_LEVELS], lcstride[YMI_MAX_LEVELS];
    float stride[YMI_MAX_LEVELS];
    float anchors[YMI_MAX_LEVELS][6]; /* 3 anchors x (w,h) in pixels */
    int32_t num_levels, n, num_classes;   /* n: images of the batch, at most 1024 (the ranking kernel's per-block table; larger batches are refused, not truncated) */
    float score_thresh, nms_thresh;
    int32_t detections_per_img;
    /* per-image rescale (transform.py:358-367): box = (box - pad) / gain; gain<=0 disables */
    const float* rescale; /* device (n,3) {gain, pad_x, pad_y} or NULL */
    float* out_boxes;
    float* out_scores;
    int64_t* out_labels;
    int32_t* out_count;
    int32_t* status; /* device int32[8] (ABI 5; was int32[4]): {records sorted (after the score-prefix selection), overflow bits, segments, largest raw
                      * per-image count on overflow, RAW candidates of the batch (every (anchor, class) pair above the threshold), blocks of the last
                      * kernel done, reserved x2} */
    void* ws;
    int64_t ws_bytes;
    int32_t cand_cap;
    int32_t flags; /* YMI_POST_* bits */
    /* ABI 5: optional packed WIRE SLAB (n, 6 * detections_per_img + 1) fp32, written by the top-k kernel itself next to the four output arrays: per image
     * [boxes 4K | scores K | labels K as float | count], slots past the count zeroed -- the fixed-shape format one all-gather per batch carries between
     * ranks (the reference's TensorRT wire format, relay/trt_graphsurgeon.py:223-244, in one buffer).  When the batch has to be re-run by the caller
     * (status[1] != 0: candidate capacity / score prefix) the LAST block of the kernel overwrites every row's count column with -1 (YMI_SLAB_STALE), so
     * the receivers of a collective enqueued right behind the post-process learn it from the data.  NULL: not written. */
    float* out_slab;
} ymi_post_desc;
#define YMI_SLAB_STALE (-1.0f)

/* ymi_post_desc.flags.
 * By default an image with many candidates is post-processed on a score-ordered PREFIX of them (at least
 * max(4096, 4 * detections_per_img) records): greedy NMS decisions depend only on higher-scored boxes, so once
 * the prefix alone yields detections_per_img kept boxes the result equals the full computation exactly.  If a
 * truncated image ends with fewer kept boxes, bit 1 of status[1] is set and the caller re-runs the batch with
 * YMI_POST_EXACT_FULL (the Python host does this transparently). */
#define YMI_POST_EXACT_FULL 1
#define YMI_STATUS_OVERFLOW_CAPACITY 1 /* status[1] bit 0: candidate capacity exceeded (grow cand_cap) */
#define YMI_STATUS_PREFIX_SHORT 2      /* status[1] bit 1: prefix too short, re-run with YMI_POST_EXACT_FULL */

/* Stem convolution fed straight from planar images (fixed-size streams): when every image of the batch already is the
 * (3, H, W) canvas -- the reference's resize is then the identity and batch_images pads nothing (transform.py:53-97,
 * 297-330) -- the letterbox pass and its NHWC4 round trip are skipped.  `d` is the stem in its super-pixel form exactly
 * as for ymi_conv2d (cin 8, 6x3, stride (2,1), pad (2,1), h = H, w_in = W/2; x is ignored), imgs[i] are device pointers
 * to contiguous (3, H, W) images of dtype d->dtype (16-byte aligned, W % 8 == 0).  Output is bit-identical to
 * ymi_letterbox + ymi_conv2d.  Replaces yolort/models/darknetv6.py:81 + common.py:69-70 on that input. */
int ymi_conv_stem_planar(const ymi_conv_desc* d, const void* const* imgs, int n_imgs, void* stream);

/* The first TWO layers of the r6.0 backbone in one launch, for the same fixed-size streams: `stem` as for ymi_conv_stem_planar
 * with 32 output channels (its y is not written), `body1` = Conv(32, 64, k=3, s=2, p=1) + BN + SiLU over the stem's output
 * (its x is not read) -- yolort/models/darknetv6.py:81 and :85-86 with common.py:69-70.  The stem's output, the largest
 * activation of the network, never reaches memory.  Output is bit-identical to ymi_conv_stem_planar + ymi_conv2d. */
int ymi_stem_body1_planar(const ymi_conv_desc* stem, const ymi_conv_desc* body1, const void* const* imgs, int n_imgs, void* stream);

/* The same pair fed from the letterboxed canvas (dynamic-shape streams: transform.py:53-97 resizes and pads first): `stem` exactly as
 * for ymi_conv2d on the NHWC4 canvas (x = ymi_letterbox's output, x_cstride 8), `body1` as above.  Bit-identical to two ymi_conv2d calls. */
int ymi_stem_body1(const ymi_conv_desc* stem, const ymi_conv_desc* body1, void* stream);

/* Measurement aid (bench.py): one wave spins for `spin_us` microseconds of the constant 100 MHz clock and writes
 * {shader-clock cycles, 100 MHz ticks} to out[0..1] (device memory) -- the shader clock the chip runs at while whatever else
 * is in flight on other streams executes.  Asynchronous on `stream`. */
int ymi_clock_probe(uint64_t* out, int spin_us, void* stream);

int64_t ymi_postprocess_ws_bytes(int n, int total_anchors, int cand_cap);
int ymi_postprocess(const ymi_post_desc* d, void* stream);

/* The same post-process in three stages, for the FUSED head (below):
 *   ymi_post_begin        resets the candidate counters of the descriptor's workspace
 *   <candidate producers> ymi_conv_head_decode once per pyramid level (any order, same or ordered streams)
 *   ymi_post_finish       sort + class-aware NMS + top-k + rescale from the appended records
 * ymi_postprocess == begin + decode of the fp32 logits + finish.  logits[] / lcstride[] are ignored by
 * begin / finish / ymi_conv_head_decode; lh / lw / stride / anchors describe the levels as before. */
int ymi_post_begin(const ymi_post_desc* d, void* stream);
int ymi_post_finish(const ymi_post_desc* d, void* stream);

/* Detection head of pyramid level `level` (the biased 1x1 conv of yolort/models/box_head.py:36,74) with the
 * decode + sigmoid + multi-label threshold of box_head.py:345-360,414-418 and _utils.py:59-60 fused into the
 * convolution's epilogue: the fp32 logits never reach memory, boxes of every anchor and the candidate records go
 * straight into `post`'s workspace (bit-identical to what ymi_postprocess derives from stored logits).
 * conv: 1x1 stride 1, cin % 32 == 0, k_pad == cin, act NONE, zeros required, y ignored.  Weight rows / bias are
 * packed per anchor: anchor q's K = num_classes + 5 outputs occupy rows q*RA .. q*RA+K-1, RA = round_up(K, 32)
 * (<= 128), remaining rows zero; cout = cout_pad = 3*RA. */
int ymi_conv_head_decode(const ymi_conv_desc* conv, const ymi_post_desc* post, int level, void* stream);
/* the heads of ALL pyramid levels in one launch (convs[l] = level l, same dtype and class count): the levels are
 * independent and the coarse ones have few blocks, so one grouped launch fills the chip where three back-to-back
 * launches leave it mostly idle.  Same records as the per-level calls. */
int ymi_conv_head_decode_group(const ymi_conv_desc* convs, int n_levels, const ymi_post_desc* post, void* stream);

/* Stand-alone class-aware NMS on caller-provided candidates of ONE image (kept indices in
 * score-descending stable order, like torchvision.ops.batched_nms called at box_head.py:422).
 * boxes (n,4) fp32 xyxy, scores (n) fp32, labels (n) int32; keep_out (n) int32, count_out int32[1]. */
int64_t ymi_nms_ws_bytes(int n);
int ymi_batched_nms(const float* boxes, const float* scores, const int32_t* labels, int n, float nms_thresh,
                    int32_t* keep_out, int32_t* count_out, void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Plan: a recorded sequence of the launches above for one (arch, N, H, W, dtype), replayed with a
 * single call (and optionally captured into a hipGraph).  Replaces the Python module-by-module
 * dispatch of yolort/models/yolo.py:159-175.  The plan copies descriptors; buffers stay owned by
 * the caller and must outlive the plan.
 * ---------------------------------------------------------------------------------------- */
typedef struct ymi_plan ymi_plan;
ymi_plan* ymi_plan_create(void);
void ymi_plan_destroy(ymi_plan* p);
int ymi_plan_add_conv(ymi_plan* p, const ymi_conv_desc* d);
int ymi_plan_add_c3_fused(ymi_plan* p, const ymi_c3_desc* d);
int ymi_plan_add_spp_pool(ymi_plan* p, void* buf, int n, int h, int w, int c, int cstride, int dtype);
int ymi_plan_add_upsample2x(ymi_plan* p, const void* x, int x_cstride, int n, int h, int w, int c, void* y,
                            int y_cstride, int dtype);
int ymi_plan_add_copy_view(ymi_plan* p, const void* x, int x_cstride, int npix, int c, void* y,
                           int y_cstride, int dtype);
int ymi_plan_add_act(ymi_plan* p, void* y, int y_cstride, int npix, int c, int dtype, int act, const void* res, int res_cstride);
int ymi_plan_add_postprocess(ymi_plan* p, const ymi_post_desc* d);
int ymi_plan_add_post_begin(ymi_plan* p, const ymi_post_desc* d);
int ymi_plan_add_head_decode(ymi_plan* p, const ymi_conv_desc* conv, const ymi_post_desc* d, int level);
int ymi_plan_add_head_decode_group(ymi_plan* p, const ymi_conv_desc* convs, int n_levels, const ymi_post_desc* d);
int ymi_plan_add_post_finish(ymi_plan* p, const ymi_post_desc* d);
int ymi_plan_num_ops(const ymi_plan* p);
/* on != 0: ops 0 and 1 (the stem over the letterboxed canvas and body.1 reading its output, see ymi_stem_body1) run as ONE launch whenever a
 * run covers both; op indices are unchanged, the stem's output buffer is then not written.  Fails unless ops 0 and 1 are such a pair. */
int ymi_plan_set_fuse_stem(ymi_plan* p, int on);
/* runs ops [first, last) on stream (last < 0 = all); use_graph != 0 replays a captured hipGraph */
int ymi_plan_run(ymi_plan* p, int first, int last, int use_graph, void* stream);
/* ---- one batch of a serving loop in two calls (the host side of yolo.py:141-183 per batch: YOLO._acquire / _submit_entry) ----
 * ymi_plan_begin: the next batch's inputs were produced on `caller_stream`; `main_stream` (the stream this plan instance launches its conv stack on) waits for
 *   them and for the plan's previous batch to have drained (its buffers are about to be overwritten).  No host synchronisation.
 * ymi_plan_submit: ops [first, n_conv) on `main_stream` (hipGraph replay when use_graph & 1), then ops [n_conv, end) -- the post-process -- on `side_stream`
 *   behind them (a second captured graph when use_graph & 2; a plan keeps two captured ranges), then `result_bytes` of `dev_result` copied to the PINNED `host_result` on `side_stream`, then the plan's completion event.  main_waits_done != 0
 *   additionally makes `main_stream` wait for that event (one batch in flight).  Neither call synchronises the host or allocates after the first use.
 * ymi_plan_done_query: 1 = the last submitted batch has completed (or none was submitted), 0 = still running.  ymi_plan_done_sync blocks the host on it. */
int ymi_plan_begin(ymi_plan* p, void* caller_stream, void* main_stream);
int ymi_plan_submit(ymi_plan* p, int first, int n_conv, int use_graph, void* main_stream, void* side_stream, const void* dev_result, void* host_result,
                    size_t result_bytes, int main_waits_done);
int ymi_plan_done_query(ymi_plan* p);
int ymi_plan_done_sync(ymi_plan* p);

/* ------------------------------------------------------------------------------------------
 * ABI 6 -- plan export / import: a recorded plan as a self-contained file, for consumers without Python (the role the
 * reference's TorchScript / ONNX artefacts of yolort/relay + yolort/runtime play).
 *   ymi_plan_export   `regions` lists every device allocation the plan's descriptors point into (base, bytes):
 *                     YMI_REGION_CONST  contents are saved (packed weights, biases, im2col tables, weight streams)
 *                     YMI_REGION_SCRATCH zero-filled at import (activation buffers with their zero tails, workspaces)
 *                     YMI_REGION_IO     zero-filled at import; the consumer finds it by `tag` (YMI_TAG_*)
 *                     Every non-NULL pointer of every recorded op must lie inside one region (else YMI_EINVAL, the op and
 *                     field named in ymi_last_error).  Synchronises `stream` and reads the constant regions back.
 *   ymi_plan_import   allocates the regions (hipMalloc), uploads the constants, rebuilds the ops; the plan owns the
 *                     memory (ymi_plan_destroy frees it).  regions_out[0 .. min(n, max_regions)) receives the new
 *                     bases with the sizes / kinds / tags of the file.  Run with ymi_plan_run as usual.
 * The file is tied to the ABI version and build it was written by (descriptor layouts, tile ids).
 * ---------------------------------------------------------------------------------------- */
#define YMI_REGION_CONST 1
#define YMI_REGION_SCRATCH 2
#define YMI_REGION_IO 3
#define YMI_TAG_NONE 0
#define YMI_TAG_INPUT 1      /* the letterboxed batch: NHWC4 canvas (n, h, w, 4) of the compute dtype, channel 3 zero (ymi_letterbox writes it) */
#define YMI_TAG_RESCALE 2    /* ymi_post_desc.rescale: fp32 (n, 3) {gain, pad_x, pad_y} per image, written by the consumer per batch (zeros: boxes stay in canvas coordinates) */
#define YMI_TAG_BOXES 3      /* fp32 (n, K, 4) */
#define YMI_TAG_SCORES 4     /* fp32 (n, K) */
#define YMI_TAG_LABELS 5     /* int64 (n, K) */
#define YMI_TAG_STATUS_COUNT 6 /* int32 [8 status words | n counts] (ymi_post_desc.status / out_count share it) */
#define YMI_TAG_SLAB 7       /* fp32 (n, 6K + 1) wire slab */
typedef struct ymi_plan_region {
    const void* base;
    int64_t bytes;
    int32_t kind, tag;
} ymi_plan_region;
int ymi_plan_export(const ymi_plan* p, const ymi_plan_region* regions, int n_regions, const char* path, void* stream);
int ymi_plan_import(const char* path, ymi_plan** out_plan, ymi_plan_region* regions_out, int max_regions, int* n_regions_out);
/* per-op timing with HIP events on `stream`: ms_out[num_ops], averaged over iters */
int ymi_plan_profile(ymi_plan* p, int iters, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLORT_AMD_H */
