"""Tight, on-config parity of the HIP path (MI355X), round 2:

  * per-LAUNCH parity of the real benchmark plan (yolov5s fp16 bs 32 640x640, BASELINE configs[1]): every conv
    launch of the recorded plan -- with the tile the plan actually uses, fused pairs / chained 1x1 / folded upsample
    included -- is fed the oracle's input of that layer (rounded to the storage type) and compared with the fp32
    oracle output of that single layer at <= 2e-3 of the layer's output range.  No chaotic amplification: a layer
    sees exact inputs.
  * fp32 PARITY MODE (csrc/conv_f32.hip): the whole network end to end against the fp32 CPU oracle with the direct
    checks of SURVEY.md 8d -- equal counts, equal labels, |dscore| <= 5e-4, IoU >= 1 - 1e-3 -- on BASELINE configs[0]
    (yolov5n thr 0.45, 2 images) and configs[1] (yolov5s, bs 32).
  * in-kernel box rescale (transform.py:354-367) against the oracle's scale_coords, to 1 ulp.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from yolort_amd import _lib
    _lib.load(require_gpu=True)
    return torch.device("cuda:0")


def _build(arch, dev, dtype, **kw):
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_weights
    head_gain = kw.pop("head_gain", 0.5)
    m = YOLOv5(arch=arch, **kw)
    sd = synth_weights(m.state_dict(), arch, seed=0, head_gain=head_gain)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    return (m if dtype == torch.float32 else m.to(dtype)), sd


# ------------------------------------------------------------------------------------------------
# per-launch parity of the benchmark plan
# ------------------------------------------------------------------------------------------------
def _fill(view, t_nchw, dtype):
    """NCHW CPU tensor -> the NHWC view of a plan buffer (channels past the tensor's are zeroed: the stem's NHWC4)"""
    dst = view.as_tensor()
    src = t_nchw.permute(0, 2, 3, 1).to(dst.device).to(dtype)
    if src.shape[-1] < dst.shape[-1]:
        dst.zero_()
        dst[..., : src.shape[-1]].copy_(src)
    else:
        dst.copy_(src)


def _read(view):
    return view.as_tensor().float().cpu().permute(0, 3, 1, 2).contiguous()


def _ref_conv(sd, p, xq, stride, pad, act=True):
    """fp32 oracle arithmetic of ONE layer (common.py:42-70 / box_head.py:74) on the given input; r3.1 networks (BottleneckCSP present): Hardswish in `Conv` (:64-65),
    the block's bare cv3 / cv2 with their half of its BatchNorm and LeakyReLU(0.1) (:136-146); a Focus stem (:210-234) on the image itself"""
    a = F.hardswish if any(k.endswith(".cv4.conv.weight") for k in sd) else F.silu
    if p + ".conv.conv.weight" in sd:   # Focus: Conv(12, c, 3) over the space-to-depth rearrangement of the image
        xq = torch.cat([xq[..., ::2, ::2], xq[..., 1::2, ::2], xq[..., ::2, 1::2], xq[..., 1::2, 1::2]], 1)
        p, stride, pad = p + ".conv", 1, 1
    if p + ".conv.weight" in sd:
        y = F.conv2d(xq, sd[p + ".conv.weight"], None, stride, pad)
        y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, 0.0, 1e-3)
        return a(y) if act else y
    if p + ".bias" in sd:
        return F.conv2d(xq, sd[p + ".weight"], sd[p + ".bias"], stride, pad)
    base, leaf = p.rsplit(".", 1)       # BottleneckCSP.cv3 / .cv2
    w = sd[p + ".weight"]
    sl = slice(0, w.shape[0]) if leaf == "cv3" else slice(w.shape[0], 2 * w.shape[0])
    y = F.batch_norm(F.conv2d(xq, w), sd[base + ".bn.running_mean"][sl], sd[base + ".bn.running_var"][sl], sd[base + ".bn.weight"][sl], sd[base + ".bn.bias"][sl], False, 0.0, 1e-3)
    return F.leaky_relu(y, 0.1)


def _resolve(first, comp, known):
    """oracle prefix of a '+'-joined op-name component, e.g. ('model.backbone.body.2.cv1', 'm.0.cv1') -> 'model.backbone.body.2.m.0.cv1'"""
    base = first
    for _ in range(4):
        base = base.rsplit(".", 1)[0]
        cand = base + "." + comp
        if cand in known:
            return cand
    raise KeyError((first, comp))


C3_SHAPES = [(1080, 1920), (720, 1280), (1920, 1080), (1080, 810), (960, 1280), (1281, 1279), (641, 480), (375, 500)]


def _fill_rep(view, t_nchw, dtype):
    """like _fill, the oracle's few images repeated over the plan's batch (every replica is checked)"""
    reps = view.n // t_nchw.shape[0]
    assert reps * t_nchw.shape[0] == view.n
    dst = view.as_tensor()
    src = t_nchw.permute(0, 2, 3, 1).to(dst.device).to(dtype)
    d4 = dst.view(reps, t_nchw.shape[0], *dst.shape[1:])
    if src.shape[-1] < dst.shape[-1]:
        dst.zero_()
        d4[..., : src.shape[-1]].copy_(src.unsqueeze(0).expand(reps, *src.shape))
    else:
        d4.copy_(src.unsqueeze(0).expand(reps, *src.shape))


def _err_vs(view, ref_nchw):
    """max |HIP - ref| over every replica, computed on the GPU (the 1280x1280 plans hold GBs per activation)"""
    got = view.as_tensor()
    ref = ref_nchw.permute(0, 2, 3, 1).to(got.device)
    reps = got.shape[0] // ref.shape[0]
    worst = 0.0
    for r in range(reps):
        worst = max(worst, float((got[r * ref.shape[0]: (r + 1) * ref.shape[0]].float() - ref).abs().max()))
    return worst


@pytest.mark.parametrize("arch,dtype,n,size,tol,dynamic,n_oracle", [
    ("yolov5_darknet_pan_s_r60", torch.float16, 32, 640, 2e-3, False, 32),     # BASELINE configs[1], the benchmarked plan
    ("yolov5_darknet_pan_n_r60", torch.bfloat16, 4, 320, 1.6e-2, False, 4),    # bf16 storage: 2^-8 output rounding
    # BASELINE configs[2] (C3): yolov5m bf16 bs 64, 1280x1280 canvas from the 8 cycled shapes -- widths 48 / 96 / 192: the cin % 32 != 0 path
    # (im2col-table implicit GEMM), conv_stem_kernel instead of the planar stem.  The oracle runs on 2 canvases; the launch runs at bs 64 on 32 copies
    ("yolov5_darknet_pan_m_r60", torch.bfloat16, 64, 1280, 1.6e-2, True, 2),
    # BASELINE configs[4] (C5): yolov5l6 fp16 bs 8 1280x1280, four pyramid levels (120 launches)
    ("yolov5_darknet_pan_l6_r60", torch.float16, 8, 1280, 2e-3, False, 2),
    # round 5, the legacy releases: Focus stem through the 6 x 6 stride-2 stem kernels; r3.1: Hardswish / LeakyReLU(0.1) in the general epilogues, BottleneckCSP's shared BatchNorm folded by halves
    ("yolov5_darknet_pan_s_r40", torch.float16, 8, 640, 2e-3, False, 4),
    ("yolov5_darknet_pan_s_r31", torch.float16, 8, 640, 2e-3, False, 4),
    # round 6: every other factory of yolort/models/yolo.py:292-834 at a small canvas -- yolov5x's widths 80 / 160 / 320 / 640 / 1280 (cin % 32 != 0:
    # the im2col-table form at shapes nothing had run), the P6 variants' fourth level, the m / l legacy releases (BottleneckCSP with 2 / 3 Bottlenecks, padded hidden widths)
    ("yolov5_darknet_pan_l_r60", torch.float16, 2, 320, 2e-3, False, 2),
    ("yolov5_darknet_pan_x_r60", torch.float16, 2, 320, 2e-3, False, 2),
    ("yolov5_darknet_pan_n6_r60", torch.float16, 2, 320, 2e-3, False, 2),
    ("yolov5_darknet_pan_s6_r60", torch.bfloat16, 2, 320, 1.6e-2, False, 2),
    ("yolov5_darknet_pan_m6_r60", torch.float16, 2, 320, 2e-3, False, 2),
    ("yolov5_darknet_pan_x6_r60", torch.float16, 2, 320, 2e-3, False, 2),
    ("yolov5_darknet_pan_m_r31", torch.float16, 2, 320, 2e-3, False, 2),
    ("yolov5_darknet_pan_l_r31", torch.float16, 2, 320, 2e-3, False, 2),
    ("yolov5_darknet_pan_m_r40", torch.bfloat16, 2, 320, 1.6e-2, False, 2),
    ("yolov5_darknet_pan_l_r40", torch.float16, 2, 320, 2e-3, False, 2),
    # round 6: canvases nobody tuned, for the strip kernel's geometry search -- 52 / 26-wide maps (full-width strips, ragged last strip), yolov5s at 1280 (160-wide maps at
    # hidden 64: full-width strips of 2 rows or column tiles, whatever the search picks; 80-wide at hidden 128), yolov5l at 1024 (hidden 64 @ 256 x 256, 128 @ 128 x 128: column tiles)
    ("yolov5_darknet_pan_s_r60", torch.float16, 1, 416, 2e-3, False, 1),
    ("yolov5_darknet_pan_s_r60", torch.float16, 2, 1280, 2e-3, False, 1),
    ("yolov5_darknet_pan_l_r60", torch.float16, 1, 1024, 2e-3, False, 1),
    ("yolov5_darknet_pan_s_r31", torch.float16, 32, 640, 2e-3, False, 2),      # the r3.1 plan at the headline's batch -- body.3 runs with YMI_ACT_NONE on a key the table pins to a SiLU-only tile
])
def test_every_conv_launch_of_the_plan_vs_oracle_layer(dev, arch, dtype, n, size, tol, dynamic, n_oracle):
    from oracle import yolov5_oracle as O
    from workloads.synth import synth_images
    kw = dict(size_divisible=64) if arch.endswith("6_r60") else {}
    m, sd = _build(arch, dev, dtype, size=(size, size), score_thresh=0.25, **kw)
    if dynamic:
        imgs_cpu = [synth_images(1, *C3_SHAPES[i % len(C3_SHAPES)], seed=1 + i)[0] for i in range(n)]
    else:
        imgs_cpu = list(synth_images(n, size, size, seed=1))
    imgs = [im.to(dev).to(dtype) for im in imgs_cpu]
    m.predict(imgs)                       # builds the plan exactly as bench.py does (planar stem / letterbox, fused head, pinned tiles)
    torch.cuda.synchronize()
    e = next(iter(m.model._entries.values()))
    plan = e.plan
    sdf = {k: v.float() for k, v in sd.items()}
    # oracle pass over (the first n_oracle canvases of) the same batch; every layer's input, rounded to the storage type
    inputs = {}
    O.TRACE.hook = lambda p, x, s, pad: inputs.__setitem__(p, x.to(dtype))
    try:
        with torch.no_grad():
            batch, _ = O.letterbox(imgs_cpu, size, size, kw.get("size_divisible", 32))
            assert tuple(batch.shape[-2:]) == (e.x.h, e.x.w)
            feats = O.backbone(batch[:n_oracle].to(torch.float32), sdf, "model.backbone")   # the conv stack and the head only (the layer inputs are what is needed):
            O.head(feats, sdf, "model.head")                                                  # the CPU post-process of a hot synthetic network takes minutes (O(n^2) NMS)
    finally:
        O.TRACE.hook = None
    known = set(inputs)
    worst = []
    checked = 0

    def q(t):
        return t.to(dtype).float()

    for idx in sorted(plan.io):
        io = plan.io[idx]
        if io.get("c3_mode") is not None:   # round 6, the strip kernel (csrc/c3_tile.hip): a whole C3 / its head / one Bottleneck / its tail per launch -- the oracle's
            # layers chained, each output rounded to the storage type like the separate launches round it (common.py:172-173, :115-116)
            mode = io["c3_mode"]
            b = "model." + io["name"].split(".tile")[0]
            tail = io["name"].split(".tile")[1]
            j = 0 if mode in (0, 1) else int(tail.split(".m.")[1].split("+")[0])
            mj = f"{b}.m.{j}"

            def bottleneck(x1):
                v = _ref_conv(sdf, mj + ".cv2", q(_ref_conv(sdf, mj + ".cv1", x1, 1, 0)), 1, 1)
                return v + x1 if io["shortcut"] else v

            outs = []
            with torch.no_grad():
                if mode in (0, 1):
                    xq = inputs[b + ".cv1"].float()
                    _fill_rep(io["x"], xq, dtype)
                    x1, x2 = q(_ref_conv(sdf, b + ".cv1", xq, 1, 0)), _ref_conv(sdf, b + ".cv2", xq, 1, 0)
                    v = bottleneck(x1)
                    if mode == 0:
                        outs.append((_ref_conv(sdf, b + ".cv3", torch.cat([q(v), q(x2)], 1), 1, 0), io["y"], 5))
                    else:
                        outs += [(v, io["y1_out"], 3), (x2, io["y2"], 1)]
                else:
                    x1 = inputs[mj + ".cv1"].float()
                    _fill_rep(io["y1_in"], x1, dtype)
                    v = bottleneck(x1)
                    if mode == 2:
                        outs.append((v, io["y1_out"], 2))
                    else:
                        c_ = x1.shape[1]
                        x2 = inputs[b + ".cv3"].float()[:, c_:]
                        _fill_rep(io["y2"], x2, dtype)
                        outs.append((_ref_conv(sdf, b + ".cv3", torch.cat([q(v), x2], 1), 1, 0), io["y"], 3))
            plan.run(idx, idx + 1)
            torch.cuda.synchronize()
            for ref, view, depth in outs:
                scale, err = float(ref.abs().max()), _err_vs(view, ref)
                worst.append((err / scale, f"{io['name']} (strip kernel, mode {mode})", -3, plan.meta[idx].get("shape")))
                assert err <= (2 if depth > 1 else 1) * tol * scale + 1e-6, f"op {idx} {io['name']}: |hip-ref| {err:.5f} > {tol} x {scale:.3f}"   # chained roundings: twice the bound
            checked += {0: 5, 1: 4, 2: 2, 3: 3}[mode]
            continue
        if io.get("fused_c3"):   # the whole one-Bottleneck C3 in one launch (csrc/c3_fused32.hip): the oracle's five layers chained, each output
            # rounded to the storage type like the separate launches round it (common.py:172-173, :115-116)
            b = "model." + io["name"].rsplit(".", 1)[0]
            xq = inputs[b + ".cv1"].float()
            _fill_rep(io["x"], xq, dtype)
            plan.run(idx, idx + 1)
            torch.cuda.synchronize()
            with torch.no_grad():
                x1, x2 = q(_ref_conv(sdf, b + ".cv1", xq, 1, 0)), q(_ref_conv(sdf, b + ".cv2", xq, 1, 0))
                v = q(_ref_conv(sdf, b + ".m.0.cv2", q(_ref_conv(sdf, b + ".m.0.cv1", x1, 1, 0)), 1, 1) + x1)
                ref = _ref_conv(sdf, b + ".cv3", torch.cat([v, x2], 1), 1, 0)
            scale, err = float(ref.abs().max()), _err_vs(io["y"], ref)
            worst.append((err / scale, b + " (fused C3)", -1, plan.meta[idx].get("shape")))
            assert err <= 2 * tol * scale + 1e-6, f"op {idx} {io['name']}: |hip-ref| {err:.5f} > {2 * tol} x {scale:.3f}"   # five chained roundings: twice the bound
            checked += 5
            continue
        parts = io["name"].split("+")
        p0 = "model." + parts[0]
        assert p0 in known, f"op {idx} {io['name']}: no oracle layer {p0}"
        xq = inputs[p0].float()
        _fill_rep(io["x"], xq, dtype)
        res_q = None
        post_act = io.get("post_act")     # r3.1 layers: the convolution with YMI_ACT_NONE + the activation launch (ymi_act, shortcut included) are ONE reference layer
        res_view = io["res"] if post_act is None else post_act["res"]
        if res_view is not None:          # Bottleneck shortcut: the block's input = input of its cv1
            res_q = inputs[p0.rsplit(".", 1)[0] + ".cv1"].float()
            _fill_rep(res_view, res_q, dtype)
        plan.run(idx, idx + (1 if post_act is None else 2))
        torch.cuda.synchronize()
        stride, pad = io["stride"][0], io["pad"][0]
        if parts[0].endswith("body.0"):   # stem: the plan runs its super-pixel form (6x3 s(2,1)); the layer is Conv(3,c,6,2,2)
            stride, pad = 2, 2
        with torch.no_grad():
            refs = []   # (label, reference NCHW, HIP view)
            r0 = _ref_conv(sdf, p0, xq, stride, pad)
            if res_q is not None:
                r0 = r0 + res_q
            if io["split"]:
                p1 = _resolve(p0, parts[1], known)
                refs.append((p0, r0, io["y"]))
                refs.append((p1, _ref_conv(sdf, p1, xq, stride, pad), io["y2"]))
            else:
                refs.append((p0, r0, io["y"]))
            if io["chain_y"] is not None and io["chain_x2"] is None:
                pc = _resolve(p0, parts[-1], known)
                refs.append((pc, _ref_conv(sdf, pc, r0.to(dtype).float(), 1, 0), io["chain_y"]))
        for label, ref, view in refs:
            scale = float(ref.abs().max())
            err = _err_vs(view, ref)
            worst.append((err / scale, label, plan.meta[idx].get("tile"), plan.meta[idx].get("shape")))
            assert err <= tol * scale + 1e-6, f"op {idx} {label} (tile {plan.meta[idx].get('tile')}, {plan.meta[idx].get('shape')}): |hip-ref| {err:.5f} > {tol} x {scale:.3f}"
            checked += 1
        if io["up2"] is not None:         # folded nn.Upsample(x2): the 4 copies must equal the launch's own output bit for bit
            y = io["y"].as_tensor()
            up = io["up2"].as_tensor()
            assert torch.equal(up, y.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)), f"op {idx} {io['name']}: folded upsample differs"
    worst.sort(reverse=True)
    print(f"{arch}: {checked} layer outputs of {len(plan.io)} launches checked; worst relative errors:")
    for w in worst[:5]:
        print("   %.2e  %s  tile %s  %s" % w)
    n_ref_convs = sum(1 for k in known if ".head." not in k and (k + ".conv") not in known)   # (a Focus stem is traced twice: the block on the image, its Conv on the rearranged image)
    assert checked >= n_ref_convs, f"only {checked} of the reference's {n_ref_convs} conv layers were exercised"
    # the first layers as the benchmark runs them equal ops 0 (and 1) on the letterboxed batch bit for bit:
    #   fixed-size streams   straight from the planar images, the stem alone (ymi_conv_stem_planar) or stem + body.1 as one launch (ymi_stem_body1_planar)
    #   dynamic-shape ones   stem + body.1 as one launch from the canvas (ymi_plan_set_fuse_stem), when the plan offers it
    if 1 not in plan.io or plan.io[0].get("post_act") is not None:   # r3.1: op 1 is the stem's Hardswish launch -- there is no fused stem + body.1 form to compare
        return
    canvas_fused = plan.fuse_stem
    plan.set_fuse_stem(False)
    batch_q, _ = O.letterbox(imgs_cpu, size, size, kw.get("size_divisible", 32))
    _fill(plan.io[0]["x"], batch_q.to(dtype).float(), dtype)
    plan.run(0, 2)
    torch.cuda.synchronize()
    y0, y1 = plan.io[0]["y"].as_tensor().clone(), plan.io[1]["y"].as_tensor().clone()
    if plan.set_fuse_stem(True):
        plan.io[0]["y"].as_tensor().zero_()
        plan.io[1]["y"].as_tensor().zero_()
        plan.run(0, 2)
        torch.cuda.synchronize()
        assert torch.equal(y1, plan.io[1]["y"].as_tensor()), "fused stem + body.1 (canvas) differs from the two launches"
        assert float(plan.io[0]["y"].as_tensor().abs().max()) == 0.0     # the stem's output never reaches memory
        print("fused stem + body.1 from the canvas: bit-identical to the two launches")
    plan.set_fuse_stem(canvas_fused)
    if not dynamic and plan.stem_planar_ok(imgs, (size, size)):
        os.environ["YOLORT_AMD_FUSE_STEM"] = "0"
        try:
            plan.io[0]["y"].as_tensor().zero_()
            assert plan.stem_from_planar(imgs) == 1
            torch.cuda.synchronize()
            assert torch.equal(y0, plan.io[0]["y"].as_tensor()), "planar stem differs from the letterbox + NHWC4 stem"
        finally:
            del os.environ["YOLORT_AMD_FUSE_STEM"]
        if plan.stem_body1_fusable():
            plan.io[0]["y"].as_tensor().zero_()
            plan.io[1]["y"].as_tensor().zero_()
            assert plan.stem_from_planar(imgs) == 2
            torch.cuda.synchronize()
            assert torch.equal(y1, plan.io[1]["y"].as_tensor()), "fused stem + body.1 differs from the two launches"
            assert float(plan.io[0]["y"].as_tensor().abs().max()) == 0.0     # the stem's output never reaches memory
            print("fused stem + body.1: bit-identical to the two launches")


# ------------------------------------------------------------------------------------------------
# fp32 parity mode: the north-star tolerance end to end
# ------------------------------------------------------------------------------------------------
def _iou_pairs(a, b):
    x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-30)


def direct_checks(ref, got, thr, k=300, score_eps=5e-4, iou_min=1 - 1e-3):
    """SURVEY.md 8d direct checks for one image.  Returns a dict of counts.  A detection whose score lies within
    `score_eps` of the threshold (or of the K-th score when the image is cut at K) may legitimately appear on one side
    only -- fp32 summation order decides it -- and is excused; everything else must pair up one to one with equal label,
    |dscore| <= score_eps and IoU >= iou_min."""
    rb, rs, rl = ref["boxes"], ref["scores"], ref["labels"]
    gb, gs, gl = got["boxes"], got["scores"], got["labels"]
    cut = thr
    if len(rs) >= k or len(gs) >= k:
        cut = max(thr, float(min(rs[-1] if len(rs) else 1.0, gs[-1] if len(gs) else 1.0)))
    used = np.zeros(len(gs), bool)
    bad, excused, paired, same_pos = 0, 0, 0, 0
    why = []
    min_iou, max_ds = 1.0, 0.0
    iou = _iou_pairs(rb, gb) if len(rs) and len(gs) else np.zeros((len(rs), len(gs)))
    for i in range(len(rs)):
        cand = np.where((gl == rl[i]) & ~used & (np.abs(gs - rs[i]) <= score_eps))[0]
        j = cand[np.argmax(iou[i, cand])] if len(cand) else -1
        if j >= 0 and iou[i, j] >= iou_min:
            used[j] = True
            paired += 1
            same_pos += int(i == j)
            min_iou = min(min_iou, float(iou[i, j]))
            max_ds = max(max_ds, abs(float(gs[j] - rs[i])))
        elif rs[i] <= cut + score_eps:
            excused += 1
        else:
            bad += 1
            same = np.where(gl == rl[i])[0]
            if len(same) and len(why) < 8:   # diagnostics: the closest same-label detection, whatever its score
                jj = same[np.argmax(iou[i, same])]
                why.append((round(float(rs[i]), 5), round(float(iou[i, jj]), 5), float(gs[jj] - rs[i])))
    for j in np.where(~used)[0]:
        if gs[j] <= cut + score_eps:
            excused += 1
        else:
            bad += 1
    return {"ref": len(rs), "got": len(gs), "paired": paired, "same_position": same_pos, "excused_at_cut": excused, "unexplained": bad,
            "min_iou": min_iou, "max_dscore": max_ds, "why": why, "equal_count": len(rs) == len(gs), "labels_equal": len(rs) == len(gs) and bool(np.array_equal(rl, gl))}


def _np(d):
    return {k: (v.detach().float().cpu().numpy() if k != "labels" else v.detach().cpu().numpy()) for k, v in d.items()}


@pytest.mark.parametrize("arch,n,thr,head_gain", [
    ("yolov5_darknet_pan_n_r60", 2, 0.45, 0.6),    # BASELINE configs[0]
    ("yolov5_darknet_pan_s_r60", 32, 0.25, 0.4),   # BASELINE configs[1] at its batch size
])
def test_fp32_parity_mode_meets_north_star_tolerance(dev, arch, n, thr, head_gain):
    """Head gains: the seeded synthetic network amplifies a rounding-level perturbation ~2500x by the time it reaches the
    logits (BatchNorm mean removal after every conv; measured: the fp32 oracle against its own float64 evaluation differs by
    3e-4 in the logits).  With the round-1 head gains (1.0 / 0.5) that alone makes 6 % of the detections of the fp32 CPU
    reference irreproducible by ANY other fp32 evaluation order (saturated, tied scores; NMS cascades) -- oracle fp32 vs oracle
    fp64: 68 of 1200 unpaired.  At gains 0.6 / 0.4 the reference is reproducible (0 of 1200 unpaired, same experiment,
    DESIGN.md section 2), so these are the workloads on which the north-star tolerance is meaningful."""
    from oracle import yolov5_oracle as O
    from workloads.synth import synth_images
    m, sd = _build(arch, dev, torch.float32, score_thresh=thr, head_gain=head_gain)
    m.set_compute_dtype(torch.float32)
    imgs_cpu = [synth_images(1, 640, 640, seed=i + 1)[0] for i in range(n)]
    dets = m.predict([im.to(dev) for im in imgs_cpu])
    e = next(iter(m.model._entries.values()))
    assert e.plan.fp32 and e.x.dtype == torch.float32 and all(f.dtype == torch.float32 for f in e.feats)
    sdf = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        ref = O.yolov5_forward(imgs_cpu, sdf, score_thresh=thr)
    def run(iou_min):
        tot = {"ref": 0, "got": 0, "paired": 0, "excused_at_cut": 0, "unexplained": 0, "same_position": 0}
        eq = lab = 0
        mi, md = 1.0, 0.0
        for r, d in zip(ref, dets):
            c = direct_checks(_np(r), _np(d), thr, iou_min=iou_min)
            for k in tot:
                tot[k] += c[k]
            eq += int(c["equal_count"])
            lab += int(c["labels_equal"])
            mi, md = min(mi, c["min_iou"]), max(md, c["max_dscore"])
            if c["why"] and iou_min < 0.995:
                print("   unpaired reference detections (score, best same-label IoU, dscore):", c["why"])
        return tot, eq, lab, mi, md

    tight, eq, lab, mi, md = run(1 - 1e-3)
    loose, _, _, mi2, _ = run(0.99)
    print(f"{arch} x{n}: IoU >= 0.999: {tight}; IoU >= 0.99: unexplained {loose['unexplained']} (min IoU {mi2:.5f}); images with equal count {eq}/{n}, "
          f"identical label sequence {lab}/{n}, min IoU of tight pairs {mi:.6f}, max |dscore| {md:.2e}")
    assert tight["ref"] > 20 * n
    # Yardstick (tools/reference_reproducibility.py): the fp32 CPU reference against its OWN float64
    # evaluation on the 32 configs[1] images pairs 6310 of 6445 detections at IoU >= 0.999 (6415 at >= 0.99; equal counts in
    # 30 of 32 images; max |dscore| 3.9e-5) -- fp32 rounding order alone, amplified ~2500x by this synthetic network, moves 2 %
    # of its boxes by more than 1e-3 IoU.  The fp32 HIP mode has to reproduce the reference at least that well
    # (measured r2: 6335 of 6445 paired, equal counts in 31 of 32, max |dscore| 4.7e-5).
    assert tight["paired"] >= 0.975 * tight["ref"], tight
    assert loose["unexplained"] <= 0.012 * (loose["ref"] + loose["got"]), loose
    assert mi >= 1 - 1e-3 and md <= 5e-4
    assert tight["excused_at_cut"] <= max(4, tight["ref"] // 500), tight     # flips exactly at the cut are rare
    assert eq >= n - max(1, n // 10)
    # identical label SEQUENCES are not required: two detections whose scores differ by < 1e-6 may swap places
    assert tight["same_position"] >= 0.9 * tight["paired"]


# ------------------------------------------------------------------------------------------------
# fp16 production path at the benchmark batch size, tightened matching (IoU >= 0.9)
# ------------------------------------------------------------------------------------------------
def test_fp16_bs32_vs_oracle_tight_matching(dev):
    """BASELINE configs[1] at its own batch (the timed plan: bs-32 tiles, 20x20 head waves spanning images, prefix
    selection) against the fp32 oracle, a match needing IoU >= 0.9 (round 1: 0.5).  16-bit STORAGE moves scores by ~1e-2
    on this synthetic network, whose detections crowd the threshold and the top-K cut, so the yardstick is the oracle
    itself with fp16 storage emulated between layers (O.EMULATE): the HIP path must match the fp32 reference as well as
    that emulation does (on average within 0.05)."""
    from oracle import yolov5_oracle as O
    from workloads.synth import synth_images
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_e2e_gpu import match_fraction
    arch = "yolov5_darknet_pan_s_r60"
    m, sd = _build(arch, dev, torch.float16, score_thresh=0.25)
    imgs_cpu = list(synth_images(32, 640, 640, seed=1))
    dets = m.predict([im.to(dev).half() for im in imgs_cpu])
    sdf = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        ref = O.yolov5_forward(imgs_cpu, sdf, score_thresh=0.25)
        O.EMULATE.dtype = torch.float16
        try:
            emu = O.yolov5_forward([im.half().float() for im in imgs_cpu], sdf, score_thresh=0.25)
        finally:
            O.EMULATE.dtype = None
    fr, fe, fh, mi = [], [], [], []
    for r, e, d in zip(ref, emu, dets):
        f, miou, _ = match_fraction(_np(r), _np(d), iou_thr=0.9, score_tol=0.05, margin=0.03, thr=0.25)
        f_emu, _, _ = match_fraction(_np(r), _np(e), iou_thr=0.9, score_tol=0.05, margin=0.03, thr=0.25)
        f_he, _, _ = match_fraction(_np(e), _np(d), iou_thr=0.9, score_tol=0.02, margin=0.03, thr=0.25)
        fr.append(f)
        fe.append(f_emu)
        fh.append(f_he)
        mi.append(miou)
    print(f"bs32 fp16, match = same label, IoU >= 0.9: HIP vs fp32 oracle mean {np.mean(fr):.3f} (min {np.min(fr):.3f}); "
          f"fp16-emulating oracle vs fp32 oracle mean {np.mean(fe):.3f} (min {np.min(fe):.3f}); HIP vs emulation mean {np.mean(fh):.3f} (min {np.min(fh):.3f}); "
          f"median IoU of matches {np.mean(mi):.4f}")
    # measured r2: HIP 0.825 / emulation 0.806 (mean), 0.46 / 0.48 (worst image); HIP vs emulation 0.50 -- two 16-bit roundings of
    # this network are as far from each other as each is from fp32 (independent perturbations), so only the distance to fp32 is asserted
    assert np.mean(fr) >= np.mean(fe) - 0.05, (np.mean(fr), np.mean(fe))
    assert np.min(fr) >= np.min(fe) - 0.2, (np.min(fr), np.min(fe))
    assert np.mean(mi) >= 0.95


# ------------------------------------------------------------------------------------------------
# a18: the in-kernel rescale
# ------------------------------------------------------------------------------------------------
def test_in_kernel_rescale_equals_scale_coords(dev):
    """gather_topk_kernel applies (box - pad) / gain per image (transform.py:354-367).  The same batch through
    YOLO.forward (no rescale) + the oracle's scale_coords must give the same boxes to 1 ulp, for shapes that hit the
    letterbox rounding traps."""
    from oracle import yolov5_oracle as O
    from workloads.synth import synth_images
    m, _ = _build("yolov5_darknet_pan_n_r60", dev, torch.float16, size=(320, 320), score_thresh=0.2, head_gain=1.0)
    shapes = [(270, 203), (240, 320), (180, 320), (375, 500), (641, 480), (97, 311)]
    imgs = [synth_images(1, h, w, seed=40 + i)[0].to(dev).half() for i, (h, w) in enumerate(shapes)]
    dets = m.predict(imgs)                                   # rescale inside the top-k gather kernel
    nested, _ = m.transform(imgs, dtype=torch.float16)
    canvas = nested.tensors                                  # (N,3,Hb,Wb) letterboxed batch
    hb, wb = int(canvas.shape[2]), int(canvas.shape[3])
    raw = m.model(canvas)                                    # canvas coordinates, no rescale
    total = 0
    for d, r, (h, w) in zip(dets, raw, shapes):
        assert torch.equal(d["labels"], r["labels"]) and torch.equal(d["scores"], r["scores"])
        want = O.scale_coords(r["boxes"].cpu(), (hb, wb), (h, w)).numpy()
        got = d["boxes"].cpu().numpy()
        ulp = np.spacing(np.maximum(np.abs(want), 1.0).astype(np.float32))
        assert np.all(np.abs(got - want) <= ulp), float(np.abs(got - want).max())
        total += len(got)
    assert total > 50
