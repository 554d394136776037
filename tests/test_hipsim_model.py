"""The yolov5s conv stack (backbone + PAN: 47 launches) on the CPU simulator, THROUGH THE PRODUCT'S OWN EMITTERS.

`engine.Plan` is given a stand-in for libyolort_amd.so whose `ymi_plan_add_*` entry points EXECUTE each op at once on the hipsim build
of the kernel sources (tests/hipsim) with host buffers, so `backbone.emit(plan, x)` -- concat elimination, cv1 + cv2 in one launch, the
chained Bottleneck 1x1, the folded upsample, channel-slice views, the SPP cascade -- runs exactly as it is recorded for the GPU, only
eagerly and on the CPU.  It is test infrastructure (the product has no CPU path: `Plan(...)` itself refuses non-CUDA devices; the plan
object here is assembled by hand), and it is what lets an opt-in kernel be tried inside the real graph without a GPU:
  * the fused one-Bottleneck C3 launch (YOLORT_AMD_FUSE_C3) leaves every pyramid feature of yolov5s BIT-IDENTICAL;
  * the features agree with the oracle's fp16-storage emulation of the reference forward.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from test_hipsim_kernels import sim  # noqa: F401  (the module-scoped fixture that builds and loads the simulator library)


class _SimLib:
    """libyolort_amd.so's plan interface, executing instead of recording"""

    def __init__(self, sim_lib, real_lib, substitute=None, small_tiles=False):
        self.sim, self.real, self.n_ops, self.tiles, self.small_tiles = sim_lib, real_lib, 0, [], small_tiles
        self.substitute = substitute or {}   # tile id -> tile id (e.g. the row-transposed-store form of a tile)

    def _done(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.sim.sim_last_error().decode()}")
        self.n_ops += 1
        return self.n_ops - 1

    # tile choice among the instantiations of the simulator build (the GPU build takes them from its pinned table)
    def _tile(self, d):
        if d.cin == 8 and d.kh == 6:
            return 41
        pointwise = d.kh == 1 and d.kw == 1 and d.sh == 1 and d.sw == 1
        if d.out_dtype == 2:   # fp32 logits of the unfused head
            return 21
        if d.chain_w:
            k1 = d.cout_split if d.cout_split > 0 else d.cout
            if pointwise and d.cin <= 128:
                return 120 + k1 // 32
            return {32: 13, 64: 12}[k1]
        if self.small_tiles and d.n * d.ho * d.wo <= 256 and d.cout_pad > 32:   # few pixels: the 64-pixel tiles (half the padded work of the 128-pixel ones)
            return 27 if d.cout_pad <= 64 else 24
        return 26 if d.cout_pad <= 32 else (27 if d.cout_pad <= 64 else 21)

    def ymi_plan_create(self):
        return 1

    def ymi_plan_destroy(self, h):
        pass

    def ymi_plan_num_ops(self, h):
        return self.n_ops

    def ymi_plan_add_conv(self, h, dref):
        d = dref._obj
        if d.tile == 0:
            d.tile = self._tile(d)
        if d.out_dtype == d.dtype:
            d.tile = self.substitute.get(int(d.tile), int(d.tile))
        self.tiles.append(int(d.tile))
        return self._done(self.sim.sim_conv2d(C.byref(d)), "sim_conv2d")

    def ymi_plan_add_c3_fused(self, h, dref):
        if dref._obj.c_hidden in (64, 128):   # the strip kernel (csrc/c3_tile.hip, round 6)
            self.tiles.append(-3)
            return self._done(self.sim.sim_c3_tile(dref), "sim_c3_tile")
        self.tiles.append(-1)
        return self._done(self.sim.sim_c3_fused(dref), "sim_c3_fused")

    def ymi_c3_tile_supported(self, dref):
        return self.sim.ymi_c3_tile_supported(dref)

    def ymi_c3_blob_bytes(self, dref):
        return self.sim.ymi_c3_blob_bytes(dref)

    def ymi_c3_pack(self, dref, blob, stream):
        return self.sim.ymi_c3_pack(dref, blob, None)

    def ymi_plan_add_spp_pool(self, h, buf, n, hh, w, c, cs, dt):
        return self._done(self.sim.ymi_spp_pool(buf, n, hh, w, c, cs, dt, None), "ymi_spp_pool")

    def ymi_plan_add_upsample2x(self, h, x, xcs, n, hh, w, c, y, ycs, dt):
        return self._done(self.sim.ymi_upsample2x(x, xcs, n, hh, w, c, y, ycs, dt, None), "ymi_upsample2x")

    def ymi_plan_add_postprocess(self, h, dref):
        rc = self.sim.ymi_postprocess(dref, None)
        return self._done(rc, "ymi_postprocess")

    # the fused head: counters, ONE launch for all levels (candidates straight into the workspace), sort / NMS / top-k
    def ymi_plan_add_post_begin(self, h, dref):
        return self._done(self.sim.ymi_post_begin(dref, None), "ymi_post_begin")

    def ymi_plan_add_head_decode_group(self, h, arr, n_levels, dref):
        return self._done(self.sim.sim_conv_head_decode_group(arr, n_levels, dref), "sim_conv_head_decode_group")

    def ymi_plan_add_post_finish(self, h, dref):
        return self._done(self.sim.ymi_post_finish(dref, None), "ymi_post_finish")

    def ymi_postprocess_ws_bytes(self, *a):
        return self.sim.ymi_postprocess_ws_bytes(*a)

    def ymi_plan_add_act(self, h, y, ycs, npix, c, dt, act, res, rcs):
        return self._done(self.sim.ymi_act(y, ycs, npix, c, dt, act, res, rcs, None), "ymi_act")

    def ymi_plan_add_copy_view(self, h, x, xcs, npix, c, y, ycs, dt):
        return self._done(self.sim.ymi_copy_view(x, xcs, npix, c, y, ycs, dt, None), "ymi_copy_view")

    def ymi_conv_build_ktab(self, *a):
        return self.real.ymi_conv_build_ktab(*a)

    def ymi_conv_f32_pick_tile(self, m, cout_pad):
        return self.sim.sim_conv_f32_pick_tile(int(m), int(cout_pad))


def _sim_plan(sim_lib, dtype, fuse_c3, substitute=None, small_tiles=False, f32_v1=False, c3_tile=False):
    from yolort_amd import _lib, engine
    p = engine.Plan.__new__(engine.Plan)   # Plan.__init__ insists on an MI355X; the attributes it would set:
    p.lib = _SimLib(sim_lib, _lib.load(require_gpu=False), substitute, small_tiles)
    p.device, p.dtype, p.handle = torch.device("cpu"), dtype, C.c_void_p(1)
    p.keep, p.names, p.meta, p.bytes_allocated, p.stream = [], [], [], 0, None
    p.zeros = torch.zeros(1024, dtype=torch.uint8)
    p.conv_descs, p.io = {}, {}
    p.chain_1x1, p.chain_cv3, p.use_v1, p.fuse_c3 = True, False, False, fuse_c3
    p.autotune, p.use_tile_table, p.fp32 = False, False, dtype == torch.float32
    p.res3x3, p.fuse_stem = (0 if p.fp32 else 2), False   # tiles 132 / 133 (conv3x3_res.hip / conv3x3_rw.hip) wherever they fit, as in production
    p.rw2 = not p.fp32                                    # tile 134 (conv3x3_rw2.hip) for Conv(64, 128, 3, 2), as in production
    p.rs = False
    p.rw3 = not p.fp32                                    # tile 135 (its K-split form for cin = 128): executed inside the whole-model runs here
    p.chain_next = False
    p.c3_tile_on = c3_tile and not p.fp32   # the strip kernel for C3 blocks of 64 / 128 hidden channels (csrc/c3_tile.hip)
    if p.fp32:   # fp32 mode (engine.Plan.__init__): the pipelined fp32 tiles with the cv1 + cv2 pair and the folded upsample; f32_v1: one register-staged launch per reference conv
        p.use_v1, p.chain_1x1, p.chain_cv3, p.fuse_c3 = f32_v1, False, False, False
    return p


def _run_backbone(sim_lib, model, img, dtype, fuse_c3, substitute=None, c3_tile=False):
    plan = _sim_plan(sim_lib, dtype, fuse_c3, substitute, c3_tile=c3_tile)
    n, _, h, w = img.shape
    x = plan.alloc(n, h, w, 4, zero=True)
    x.as_tensor()[..., :3] = img.permute(0, 2, 3, 1).to(dtype)
    feats = model.model.backbone.emit(plan, x)
    out = [f.as_tensor().clone() for f in feats]
    plan.handle = None   # nothing to destroy
    return out, plan


def test_yolov5s_conv_stack_on_the_simulator_and_fused_c3_inside_it(sim):
    from oracle import yolov5_oracle as O
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch, dtype, S = "yolov5_darknet_pan_s_r60", torch.float16, 64
    model = YOLOv5(arch=arch, size=(S, S), score_thresh=0.25)
    model.load_state_dict(synth_weights(model.state_dict(), arch, seed=0, head_gain=0.4))
    model = model.to(dtype).eval()
    img = synth_images(1, S, S, seed=5)[0][None].to(dtype).float()

    sep, plan_sep = _run_backbone(sim, model, img, dtype, fuse_c3=False)
    fused, plan_fused = _run_backbone(sim, model, img, dtype, fuse_c3=True)
    assert plan_sep.num_ops == 47 and plan_fused.num_ops == 45, (plan_sep.num_ops, plan_fused.num_ops)   # the C3 at 160^2-equivalent: three launches -> one
    assert plan_fused.names[2].endswith(".fused") and plan_fused.lib.tiles[2] == -1
    for a, b in zip(sep, fused):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))

    # round 6: every C3 of 64 / 128 hidden channels through the strip kernel (csrc/c3_tile.hip) -- body.4 (two Bottlenecks: HEAD + TAIL), body.6 (three: HEAD, MID, TAIL),
    # the PAN's one-Bottleneck blocks as single launches: the pyramid features stay BIT-IDENTICAL (the separate 3x3 launches here run on tiles whose k order the
    # strip kernel does not share at every width, so identity is asserted against a run that pins them to the LDS-halo kernels' order)
    strip, plan_strip = _run_backbone(sim, model, img, dtype, fuse_c3=True, c3_tile=True)
    n_strip = sum(t == -3 for t in plan_strip.lib.tiles)
    assert n_strip >= 7 and plan_strip.num_ops < plan_fused.num_ops, (n_strip, plan_strip.num_ops)
    for a, b in zip(fused, strip):
        d = (a.float() - b.float()).abs().max().item()
        assert d <= 2e-2 * max(1.0, a.float().abs().max().item()), d

    # the row-transposed-store tiles (141 / 142 = tiles 12 / 21: StoreEpilogueTP) in place of their base tiles, everywhere in the graph --
    # plain, residual, split, upsampled-copy and chained launches included (the last three fall back to the plain stores inside the kernel)
    tp, plan_tp = _run_backbone(sim, model, img, dtype, fuse_c3=False, substitute={12: 141, 21: 142})
    assert sum(t in (141, 142) for t in plan_tp.lib.tiles) >= 15
    for a, b in zip(sep, tp):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))

    sd = {k: v.float() for k, v in model.state_dict().items()}
    O.EMULATE.dtype = dtype
    try:
        with torch.no_grad():
            ref = O.backbone(img, sd, p="model.backbone")
    finally:
        O.EMULATE.dtype = None
    assert len(ref) == len(sep) == 3
    for i, (r, g) in enumerate(zip(ref, sep)):
        got = g.float().permute(0, 3, 1, 2)
        assert got.shape == r.shape
        err, scale = (got - r).abs().max().item(), r.abs().max().item()
        print(f"feature {i}: {tuple(r.shape)} max |sim - emulation| = {err:.4g} (range {scale:.4g})")
        assert err <= 2e-2 * max(1.0, scale)


@pytest.mark.parametrize("arch,S,div,gain,dtype,fused_head", [("yolov5_darknet_pan_n_r60", 96, 32, 0.5, torch.float16, False), ("yolov5_darknet_pan_n_r60", 96, 32, 0.5, torch.float16, True),
                                                            ("yolov5_darknet_pan_l6_r60", 128, 64, 4.0, torch.float16, True), ("yolov5_darknet_pan_m_r60", 64, 32, 2.0, torch.bfloat16, False),
                                                            # the legacy releases (round 5): Focus stem as the 6 x 6 stride-2 convolution it equals; r3.1: BottleneckCSP, Hardswish / LeakyReLU(0.1) epilogues
                                                            ("yolov5_darknet_pan_s_r40", 64, 32, 0.25, torch.float16, True), ("yolov5_darknet_pan_s_r31", 64, 32, 0.25, torch.float16, True),
                                                            ("yolov5_darknet_pan_m_r31", 64, 32, 0.25, torch.float16, False), ("yolov5_darknet_pan_m_r40", 64, 32, 0.25, torch.bfloat16, False)])   # widths 48 / 96 / 192: padded hidden buffers under the activation launch
def test_yolov5n_detections_on_the_simulator_vs_oracle(sim, arch, S, div, gain, dtype, fused_head):
    """letterbox -> backbone + PAN -> head -> decode / sort / NMS / top-k, every kernel on the simulator, driven by the product's
    emitters and its host recipe; against the oracle's fp32 forward with the matching criterion of __graft_entry__.smoke().
    The head runs in its unfused form (fp32 logits + decode kernel) or -- since round 3, when its per-wave worklist got its wave fences -- as
    the shipped fused head-decode launch (all levels in one launch: what the product records); yolov5m's head inputs are not 32-aligned
    (the product takes the unfused form there too)."""
    from oracle import yolov5_oracle as O
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    from test_hipsim_kernels import _sim_letterbox
    thr = 0.1   # second case: a P6 model (IntermediateLevelP6, four pyramid levels, size_divisible = 64; yolo.py:622-834); third: yolov5m in bf16 --
    #             widths 48 / 96 / 192 ...: channel counts that are not multiples of 32 take the im2col-table form of the implicit GEMM
    model = YOLOv5(arch=arch, size=(S, S), score_thresh=thr, **(dict(size_divisible=div) if div != 32 else {}))
    model.load_state_dict(synth_weights(model.state_dict(), arch, seed=0, head_gain=gain))
    model = model.to(dtype).eval()
    imgs = [synth_images(1, 3 * S // 4, S, seed=21)[0], synth_images(1, S, 5 * S // 8, seed=22)[0]][: 2 if arch.endswith("_n_r60") else 1]   # (one image for the larger models: time)
    with torch.no_grad():
        ref = O.yolov5_forward(imgs, {k: v.float() for k, v in model.state_dict().items()}, size=(S, S), size_divisible=div, score_thresh=thr)

    canvas, sizes = _sim_letterbox(sim, [im.to(dtype) for im in imgs], S, dtype, div=div)       # (n, hb, wb, 4) NHWC4, as the plan's input view
    n, hb, wb, _ = canvas.shape
    plan = _sim_plan(sim, dtype, fuse_c3=False, small_tiles=not arch.endswith("_n_r60"))
    x = plan.alloc(n, hb, wb, 4, zero=True)
    x.as_tensor().copy_(canvas)
    yolo = model.model
    feats = yolo.backbone.emit(plan, x)
    ag = yolo.anchor_generator
    rescale = torch.zeros(n, 3, dtype=torch.float32)
    for i, im in enumerate(imgs):   # transform.py:354-367: gain and pad of each image inside the canvas
        h0, w0 = int(im.shape[-2]), int(im.shape[-1])
        gain = min(hb / h0, wb / w0)
        rescale[i] = torch.tensor([gain, (wb - w0 * gain) / 2, (hb - h0 * gain) / 2])
    strides = [float(s_) for s_ in ag.strides]
    if fused_head:   # what YOLO._build_entry records (models/yolo.py): post_begin, the fused heads of all levels in ONE launch, post_finish
        assert yolo.head.can_fuse_decode(plan, feats)
        pb, pd = plan.post_desc([(f.h, f.w) for f in feats], n, strides, ag.anchor_grids, yolo.num_classes, thr, 0.45, 300, 32768 * n, rescale=rescale)
        plan.post_begin(pd)
        yolo.head.emit_fused(plan, feats, pd)
        plan.post_finish(pd, pb.total_anchors)
    else:
        logits = yolo.head.emit(plan, feats)
        pb = plan.postprocess(logits, strides, ag.anchor_grids, yolo.num_classes, thr, 0.45, 300, 32768 * n, rescale=rescale)
    assert int(pb.status[1]) == 0, pb.status.tolist()
    plan.handle = None
    for i, r in enumerate(ref):
        c = int(pb.count[i])
        gb, gs, gl = pb.boxes[i, :c].numpy(), pb.scores[i, :c].numpy(), pb.labels[i, :c].numpy()
        rb, rs, rl = r["boxes"].numpy(), r["scores"].numpy(), r["labels"].numpy()
        nr = len(rs)
        assert nr > 0 and abs(nr - c) <= max(3, nr // 10), (nr, c)
        need = [j for j in range(nr) if rs[j] >= max(thr, float(rs.min())) + 0.03]   # references near the threshold / the top-k cut need not survive fp16 storage
        hit = 0
        for j in need:
            cand = np.where((gl == rl[j]) & (np.abs(gs - rs[j]) < 0.05))[0]
            if len(cand):
                x1, y1 = np.maximum(gb[cand, 0], rb[j, 0]), np.maximum(gb[cand, 1], rb[j, 1])
                x2, y2 = np.minimum(gb[cand, 2], rb[j, 2]), np.minimum(gb[cand, 3], rb[j, 3])
                inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
                iou = inter / ((gb[cand, 2] - gb[cand, 0]) * (gb[cand, 3] - gb[cand, 1]) + (rb[j, 2] - rb[j, 0]) * (rb[j, 3] - rb[j, 1]) - inter)
                hit += int(iou.max() >= 0.5)
        print(f"image {i}: {nr} reference detections, {c} from the simulator, {hit}/{len(need)} matched")
        assert hit >= (0.9 if dtype == torch.float16 else 0.8) * len(need)   # bf16 storage: 8 mantissa bits (the GPU suite's bf16 bound is looser still)


@pytest.mark.parametrize("f32_v1", [False, True])
def test_fp32_parity_mode_on_the_simulator_meets_the_north_star_tolerance(sim, f32_v1):
    """the fp32 mode (csrc/conv_f32_pipe.hip: pipelined fp32 tiles, cv1 + cv2 in one launch, folded upsample; f32_v1: csrc/conv_f32.hip, one launch per
    reference conv; fp32 pool / upsample / logits: `YOLOv5.set_compute_dtype(torch.float32)` on the GPU) end to end
    on the simulator, yolov5n on two differently shaped images, against the fp32 oracle with the DIRECT checks of SURVEY.md 8d:
    equal counts, equal labels, |score difference| <= 1e-4, IoU >= 1 - 1e-3"""
    from oracle import yolov5_oracle as O
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    from test_hipsim_kernels import _sim_letterbox
    sim.ymi_copy_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    arch, dtype, S, thr = "yolov5_darknet_pan_n_r60", torch.float32, 96, 0.25
    model = YOLOv5(arch=arch, size=(S, S), score_thresh=thr)
    model.load_state_dict(synth_weights(model.state_dict(), arch, seed=0, head_gain=0.5))
    model = model.float().eval()
    imgs = [synth_images(1, 72, 96, seed=21)[0], synth_images(1, 96, 60, seed=22)[0]]
    with torch.no_grad():
        ref = O.yolov5_forward(imgs, {k: v.float() for k, v in model.state_dict().items()}, size=(S, S), score_thresh=thr)
    canvas, _ = _sim_letterbox(sim, imgs, S, dtype)
    n, hb, wb, _ = canvas.shape
    plan = _sim_plan(sim, dtype, fuse_c3=False, f32_v1=f32_v1)
    x = plan.alloc(n, hb, wb, 4, zero=True)
    x.as_tensor().copy_(canvas)
    yolo = model.model
    feats = yolo.backbone.emit(plan, x)
    logits = yolo.head.emit(plan, feats)
    if not f32_v1:   # the launches the GPU plan records: pipelined tiles everywhere, the C3 pairs and both folded upsamples
        assert all(201 <= t <= 206 for t in plan.lib.tiles), plan.lib.tiles
        assert sum(n_.endswith(".cv1+cv2") for n_ in plan.names) == 8 and not any("upsample" in n_ or "inner_blocks.2" in n_ or "inner_blocks.5" in n_ for n_ in plan.names), plan.names
    ag = yolo.anchor_generator
    rescale = torch.zeros(n, 3, dtype=torch.float32)
    for i, im in enumerate(imgs):
        h0, w0 = int(im.shape[-2]), int(im.shape[-1])
        gain = min(hb / h0, wb / w0)
        rescale[i] = torch.tensor([gain, (wb - w0 * gain) / 2, (hb - h0 * gain) / 2])
    pb = plan.postprocess(logits, [float(s_) for s_ in ag.strides], ag.anchor_grids, yolo.num_classes, thr, 0.45, 300, 32768 * n, rescale=rescale)
    assert int(pb.status[1]) == 0, pb.status.tolist()
    plan.handle = None
    total = paired = 0
    for i, r in enumerate(ref):
        c = int(pb.count[i])
        gb, gs, gl = pb.boxes[i, :c].numpy(), pb.scores[i, :c].numpy(), pb.labels[i, :c].numpy()
        rb, rs, rl = r["boxes"].numpy(), r["scores"].numpy(), r["labels"].numpy()
        assert c == len(rs), (c, len(rs))
        used = np.zeros(c, bool)
        for j in range(len(rs)):   # a reference detection is paired with an unused detection of the same label, score within 1e-4, IoU >= 1 - 1e-3
            cand = np.where((gl == rl[j]) & (np.abs(gs - rs[j]) <= 1e-4) & ~used)[0]
            if not len(cand):
                continue
            x1, y1 = np.maximum(gb[cand, 0], rb[j, 0]), np.maximum(gb[cand, 1], rb[j, 1])
            x2, y2 = np.minimum(gb[cand, 2], rb[j, 2]), np.minimum(gb[cand, 3], rb[j, 3])
            inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
            iou = inter / ((gb[cand, 2] - gb[cand, 0]) * (gb[cand, 3] - gb[cand, 1]) + (rb[j, 2] - rb[j, 0]) * (rb[j, 3] - rb[j, 1]) - inter)
            if iou.max() >= 1 - 1e-3:
                used[cand[int(iou.argmax())]] = True
                paired += 1
        total += len(rs)
    print(f"fp32 parity mode on the simulator: {paired} of {total} reference detections paired (same label, |dscore| <= 1e-4, IoU >= 1 - 1e-3)")
    assert total > 50 and paired == total


@pytest.mark.parametrize("name", ["Conv.output_shape", "Conv.stride_2", "Conv.version_r31", "Conv.no_activation", "Bottleneck.with_shortcut", "Bottleneck.without_shortcut",
                                  "C3.output_shape", "C3.different_channels", "SPP.output_shape", "SPPF.output_shape", "Focus.output_shape", "Focus.r31", "BottleneckCSP.r31"])
def test_reference_block_cases_on_the_simulator(sim, name):
    """the block-level cases of the reference's test/test_v5_common.py (tests/_blocks.py: same constructor calls, same input shapes) through the product's emitters on the
    simulator: the shape the reference asserts, and the oracle's fp32 value of the same module"""
    import _blocks
    dtype = torch.float16
    m, x, out_shape = _blocks.build(name)
    want = _blocks.expected(name, m, x)
    m = m.to(dtype)
    plan = _sim_plan(sim, dtype, fuse_c3=False, small_tiles=True)
    n, c, h, w = x.shape
    xv = plan.alloc(n, h, w, m._input_cpad(c), zero=True)
    xv.as_tensor()[..., :c] = x.permute(0, 2, 3, 1).to(dtype)
    y = m.emit(plan, xv)
    plan.handle = None
    got = y.as_tensor()[..., :y.c].float().permute(0, 3, 1, 2)
    assert tuple(got.shape) == out_shape == tuple(want.shape)
    assert (got - want).abs().max().item() <= 1e-2 * max(1.0, want.abs().max().item()), name
