"""The fused one-Bottleneck C3 launch (csrc/c3_fused32.hip, ymi_c3_fused) and the row-transposed-store tiles.

Both were written blind at the end of round 2 (CPU simulator only) and gated; their first GPU run (round 3, tools/gpu_calls/gpu_r3_c3fused.sh:
22 passed) un-gated them -- the fused launch is the default for yolov5s body.2 now (YOLORT_AMD_FUSE_C3=0 restores the separate launches).

What they pin: the fused launch is BIT-IDENTICAL to the three separate launches (same rounding points, same k
order on top of the bias) on ragged sizes, channel-slice views and both 16-bit types, agrees with the fp32 torch evaluation of
common.py:172-173 / :115-116 within the per-launch tolerance, and leaves yolov5s detections unchanged end to end.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu]


@pytest.fixture(scope="module")
def dev():
    from yolort_amd import _lib
    _lib.load(require_gpu=True)
    return torch.device("cuda:0")


def _make_c3(seed):
    from yolort_amd.v5.models.common import C3
    torch.manual_seed(seed)
    m = C3(64, 64, n=1).eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.6, 1.4)
                mod.bias.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
    return m


def _torch_c3(m, x, dtype):
    """fp32 evaluation of the reference forward on operands rounded like the HIP path rounds them (weights after the BN fold,
    every layer output to the storage dtype)"""
    def conv(c, t, res=None):
        bn = c.bn
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (c.conv.weight * scale.view(-1, 1, 1, 1)).to(dtype).float()
        b = bn.bias - bn.running_mean * scale
        y = F.silu(F.conv2d(t, w, b, c.conv.stride, c.conv.padding))
        if res is not None:
            y = y + res
        return y.to(dtype).float()
    with torch.no_grad():
        x1, x2 = conv(m.cv1, x), conv(m.cv2, x)
        v = conv(m.m[0].cv2, conv(m.m[0].cv1, x1), res=x1)
        return conv(m.cv3, torch.cat([v, x2], 1))


def _run(dev, m, x, dtype, fuse, x_extra=0, y_extra=0):
    from yolort_amd import engine
    plan = engine.Plan(dev, dtype)
    plan.fuse_c3 = fuse
    n, _, h, w = x.shape
    xb = plan.alloc(n, h, w, 64 + x_extra, zero=True)
    xv = xb.slice_c(x_extra // 2, 64) if x_extra else xb
    xv.as_tensor().copy_(x.permute(0, 2, 3, 1).to(dev, dtype))
    yb = plan.alloc(n, h, w, 64 + y_extra, zero=True)
    yv = yb.slice_c(y_extra // 2, 64) if y_extra else yb
    m.emit(plan, xv, out=yv, name="c3")
    assert (plan.num_ops == 1) == fuse
    plan.run()
    torch.cuda.synchronize()
    return yv.as_tensor().cpu(), yb.as_tensor().cpu()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 16, 16), (2, 21, 37), (1, 5, 3), (3, 33, 16), (2, 160, 160), (9, 48, 80)])
def test_fused_c3_equals_the_separate_launches(dev, dtype, shape):
    n, h, w = shape
    m = _make_c3(seed=h * 7 + w)
    x = torch.randn(n, 64, h, w, generator=torch.Generator().manual_seed(n + h)).to(dtype).float()
    sep, _ = _run(dev, m, x, dtype, fuse=False)
    fused, _ = _run(dev, m, x, dtype, fuse=True)
    # bit-identical when the separate launches run on the kernels the pinned table gives this layer (streaming 1x1, resident-weights 3x3:
    # what the CPU simulator test pins); at shapes outside the table the heuristic may pick tiles that round the 3x3's partial sums in another
    # grouping, so a last-bit difference on a few elements is tolerated here and reported
    if not torch.equal(sep.view(torch.int16), fused.view(torch.int16)):
        d = (sep.float() - fused.float()).abs()
        frac = (d > 0).float().mean().item()
        print(f"fused vs separate launches at {shape} {dtype}: {frac:.2e} of the elements differ, max {d.max().item():.3g}")
        ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
        assert d.max().item() <= 2 * ulp * max(1.0, sep.float().abs().max().item()) and frac < 1e-2
    ref = _torch_c3(m, x, dtype).permute(0, 2, 3, 1)
    tol = 4e-3 if dtype == torch.float16 else 3.2e-2   # five chained layers: twice the per-launch bound of test_ops_gpu.py
    err = (fused.float() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


def test_fused_c3_on_channel_slice_views(dev):
    """input and output as 64-channel slices of wider buffers (pixel strides 96 / 128): nothing outside the slice is written"""
    m = _make_c3(seed=5)
    x = torch.randn(2, 64, 19, 23, generator=torch.Generator().manual_seed(9)).half().float()
    sep, _ = _run(dev, m, x, torch.float16, fuse=False, x_extra=32, y_extra=64)
    fused, whole = _run(dev, m, x, torch.float16, fuse=True, x_extra=32, y_extra=64)
    assert torch.equal(sep.view(torch.int16), fused.view(torch.int16))
    assert whole[..., :32].abs().max().item() == 0 and whole[..., 96:].abs().max().item() == 0


def test_yolov5s_detections_unchanged_by_the_fused_c3(dev, monkeypatch):
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_s_r60"
    imgs = [im.to(dev) for im in synth_images(2, 640, 640, seed=3)]
    outs = []
    for knob in ("0", "1"):
        monkeypatch.setenv("YOLORT_AMD_FUSE_C3", knob)
        model = YOLOv5(arch=arch, size=(640, 640), score_thresh=0.25)
        model.load_state_dict(synth_weights(model.state_dict(), arch, seed=0, head_gain=0.4))
        model = model.to(dev).half().eval()
        outs.append(model.predict(imgs))
        torch.cuda.synchronize()
    for a, b in zip(*outs):
        for k in ("scores", "labels", "boxes"):
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("tile,base,cout", [(141, 12, 64), (142, 21, 128), (143, 66, 128), (144, 61, 128), (145, 71, 128), (151, 111, 128), (152, 112, 64), (155, 115, 256)])
def test_row_transposed_store_tiles_equal_their_base_tiles(dev, tile, base, cout):
    """tiles 141-145 / 151-155 (StoreEpilogueTP: the lean epilogue's packets leave as whole rows through a wave-private LDS tile): the same
    bits as the tile they derive from.  On the CPU simulator they already are (tests/test_hipsim_kernels.py); this is their first GPU run."""
    from yolort_amd import engine

    def run(t, dtype, n, cin, h, w, k, s, residual):
        g = torch.Generator().manual_seed(tile)
        x = torch.randn(n, cin, h, w, generator=g).to(dtype).float()
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dtype).float()
        bias = torch.randn(cout, generator=g) * 0.1
        plan = engine.Plan(dev, dtype)
        xv = plan.alloc(n, h, w, cin)
        xv.as_tensor().copy_(x.permute(0, 2, 3, 1).to(dev, dtype))
        pc = engine.PackedConv(wt, bias, None, dtype, dev)
        ho, wo = engine.conv_out_hw(h, w, (k, k), (s, s), (k // 2, k // 2))
        wide = plan.alloc(n, ho, wo, cout + 64, zero=True)
        rv = None
        if residual:
            rv = plan.alloc(n, ho, wo, cout)
            rv.as_tensor().copy_(torch.randn(n, ho, wo, cout, generator=g).to(dev, dtype))
        plan.conv(xv, pc, s, k // 2, out=wide.slice_c(32, cout), res=rv, tile=t)
        plan.run()
        torch.cuda.synchronize()
        return wide.as_tensor().cpu()

    for dtype, cfg in [(torch.float16, (2, 64, 37, 29, 1, 1, True)), (torch.bfloat16, (3, 64, 40, 40, 3, 1, False)), (torch.float16, (2, 32, 33, 21, 3, 2, False)),
                       (torch.float16, (8, 128, 80, 80, 1, 1, False))]:
        a, b = run(base, dtype, *cfg), run(tile, dtype, *cfg)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), cfg


def test_fused_stem_body1_leaves_yolov5s_detections_unchanged(dev, monkeypatch):
    """stem + body.1 as ONE launch from the planar images (csrc/stem_body1_fused.hip, ymi_stem_body1_planar: the default for fixed-size yolov5s
    streams since round 3) against the two launches (YOLORT_AMD_FUSE_STEM=0): identical detections, bit for bit; the per-launch parity test
    (tests/test_parity_gpu.py) compares body.1's output itself"""
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_s_r60"
    outs = []
    for shape in ((640, 640), (352, 608)):
        imgs = [im.to(dev).half() for im in synth_images(3, *shape, seed=5)]
        res = []
        for knob in ("0", "1"):
            monkeypatch.setenv("YOLORT_AMD_FUSE_STEM", knob)
            model = YOLOv5(arch=arch, size=(640, 640), score_thresh=0.25)
            model.load_state_dict(synth_weights(model.state_dict(), arch, seed=0, head_gain=0.4))
            model = model.to(dev).half().eval()
            res.append(model.predict(imgs))
            torch.cuda.synchronize()
            e = next(iter(model.model._entries.values()))
            assert e.plan.stem_body1_fusable() == (knob == "1")
        assert sum(len(a["scores"]) for a in res[0]) > 0
        for a, b in zip(*res):
            for k in ("scores", "labels", "boxes"):
                assert torch.equal(a[k], b[k]), (shape, k)


# ---------------------------------------------------------------------------------------------------------------------
# round 6: the strip kernel (csrc/c3_tile.hip) -- C3 blocks of 64 / 128 hidden channels, one launch per block or per Bottleneck
# ---------------------------------------------------------------------------------------------------------------------
def _make_c3_wide(c1, c2, n, shortcut, seed):
    from yolort_amd.v5.models.common import C3
    torch.manual_seed(seed)
    m = C3(c1, c2, n=n, shortcut=shortcut).eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.6, 1.4)
                mod.bias.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
    return m


def _torch_c3_wide(m, x, dtype, shortcut):
    def conv(c, t, res=None):
        bn = c.bn
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (c.conv.weight * scale.view(-1, 1, 1, 1)).to(dtype).float()
        y = F.silu(F.conv2d(t, w, bn.bias - bn.running_mean * scale, c.conv.stride, c.conv.padding))
        if res is not None:
            y = y + res
        return y.to(dtype).float()
    with torch.no_grad():
        x1, x2 = conv(m.cv1, x), conv(m.cv2, x)
        for b in m.m:
            x1 = conv(b.cv2, conv(b.cv1, x1), res=x1 if shortcut else None)
        return conv(m.cv3, torch.cat([x1, x2], 1))


def _run_strip(dev, m, x, dtype, strip, t3x3):
    """strip = True: C3.emit with the strip kernel; False: the separate launches recorded by hand on tiles whose k order the strip kernel shares (1x1: tile 21; 3x3: the
    8-wave LDS-halo kernel) -- what the comparison is bit for bit against"""
    from yolort_amd import engine
    plan = engine.Plan(dev, dtype)
    n, c1, h, w = x.shape
    c_ = m.cv1.conv.out_channels
    nb = len(m.m)
    xv = plan.alloc(n, h, w, c1)
    xv.as_tensor().copy_(x.permute(0, 2, 3, 1).to(dev, dtype))
    wide = plan.alloc(n, h, w, 2 * c_ + 64, zero=True)
    out = wide.slice_c(32, 2 * c_)
    if strip:
        plan.c3_tile_on = True
        m.emit(plan, xv, out=out, name="c3")
        assert plan.num_ops == max(1, nb), plan.names
    else:
        pk = lambda cv, cin: cv.packed(dtype, dev, cin)  # noqa: E731
        cat = plan.alloc(n, h, w, 2 * c_)
        y = plan.alloc(n, h, w, c_)
        plan.conv(xv, m.packed_pair(dtype, dev, c1), out=y, out2=cat.slice_c(c_, c_), split=c_, tile=21)
        for j, b in enumerate(m.m):
            t = plan.conv(y, pk(b.cv1, c_), tile=21)
            y = plan.conv(t, pk(b.cv2, c_), 1, 1, out=cat.slice_c(0, c_) if j == nb - 1 else None, res=y if b.add else None, tile=t3x3)
        plan.conv(cat, pk(m.cv3, 2 * c_), out=out, tile=21)
    plan.run()
    torch.cuda.synchronize()
    return out.as_tensor().cpu(), wide.as_tensor().cpu()


@pytest.mark.parametrize("case", [
    # (dtype, n, h, w, c_in, hidden, bottlenecks, shortcut)
    (torch.float16, 32, 40, 40, 256, 128, 1, False),   # yolov5s pan.layer_blocks.2 at the benchmark's batch
    (torch.float16, 32, 40, 40, 512, 128, 1, False),   # pan.inner_blocks.3
    (torch.float16, 32, 40, 40, 256, 128, 3, True),    # backbone.body.6: HEAD, MID, TAIL
    (torch.float16, 32, 80, 80, 128, 64, 2, True),     # backbone.body.4: HEAD, TAIL
    (torch.float16, 32, 80, 80, 256, 64, 1, False),    # pan.layer_blocks.0
    (torch.bfloat16, 3, 37, 40, 96, 128, 2, True),     # ragged last strip
    (torch.bfloat16, 2, 50, 76, 64, 64, 1, True),
    (torch.float16, 5, 20, 20, 128, 128, 1, False),
    (torch.float16, 300, 13, 24, 64, 64, 1, False),    # more strips than blocks: the ring runs across strips
    (torch.float16, 8, 320, 320, 128, 64, 3, True),    # yolov5l6 @ 1280 backbone.body.2: wider than the patch -> column tiles (HEAD, MID, TAIL)
    (torch.float16, 8, 160, 160, 256, 128, 2, True),   # ... backbone.body.4's first two Bottlenecks: hidden 128 on a 160-wide map
    (torch.bfloat16, 3, 45, 203, 64, 128, 1, False),   # ragged rows and columns
])
def test_c3_strip_kernel_equals_the_separate_launches(dev, case):
    dtype, n, h, w, c1, c_, nb, shortcut = case
    m = _make_c3_wide(c1, 2 * c_, nb, shortcut, seed=h * 7 + w + nb)
    x = torch.randn(n, c1, h, w, generator=torch.Generator().manual_seed(n + h)).to(dtype).float()
    sep, _ = _run_strip(dev, m, x, dtype, False, 91 if c_ == 128 else 92)
    got, whole = _run_strip(dev, m, x, dtype, True, 0)
    assert torch.equal(sep.view(torch.int16), got.view(torch.int16)), f"max difference {(sep.float() - got.float()).abs().max().item()}"
    assert whole[..., :32].abs().max().item() == 0 and whole[..., 32 + 2 * c_:].abs().max().item() == 0
    ref = _torch_c3_wide(m, x[: min(n, 4)], dtype, shortcut).permute(0, 2, 3, 1)
    tol = (4e-3 if dtype == torch.float16 else 3.2e-2) * (1 + nb) / 2
    err = (got[: min(n, 4)].float() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("arch,n,h,w,geom,blocks", [
    ("yolov5_darknet_pan_s_r60", 3, 544, 960, "3,2,2", 40),      # non-square canvas: 68 x 120 / 34 x 60 maps, ragged rows and columns
    ("yolov5_darknet_pan_l6_r60", 1, 768, 1024, "2,1,5", 64),    # four levels; hidden 64 @ 192 x 256, hidden 128 @ 96 x 128: column tiles either way
    ("yolov5_darknet_pan_n_r60", 5, 352, 416, "1,1,1", 3),       # yolov5n: hidden 64 at 44 x 52, 128 at 22 x 26; one row per strip, three blocks
])
def test_strip_geometry_never_changes_a_detection_on_other_models_and_canvases(dev, arch, n, h, w, geom, blocks, monkeypatch):
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    imgs = [im.to(dev).half() for im in synth_images(n, h, w, seed=h + n)]
    kw = dict(size_divisible=64) if arch.endswith("6_r60") else {}
    outs, strips = [], []
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("YOLORT_AMD_C3T_GEOM", geom)
            monkeypatch.setenv("YOLORT_AMD_C3T_BLOCKS", str(blocks))
        m = YOLOv5(arch=arch, size=(h, w), score_thresh=0.1, **kw)
        m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=0.8))
        m = m.to(dev).half().eval()
        outs.append(m(imgs))
        e = next(iter(m.model._entries.values()))
        strips.append(sum(".tile" in nm for nm in e.plan.names))
    assert strips[0] == strips[1] and strips[0] >= 2, strips   # (a forced geometry the kernel cannot run would silently fall back to the separate launches: not here)
    assert sum(len(d["scores"]) for d in outs[0]) > 5
    for a, b in zip(*outs):
        for k in ("boxes", "scores", "labels"):
            assert torch.equal(a[k], b[k]), (arch, geom, k)


@pytest.mark.parametrize("geom,blocks", [("4,1,2", 50), ("3,5,3", 7), ("2,9,1", 300)])
def test_strip_geometry_never_changes_a_detection(dev, geom, blocks, monkeypatch):
    """a strip's arithmetic does not depend on how the map is cut: whole yolov5s, the library's geometry against a forced one ("rows per strip, slot origin, column tiles")
    walked by few blocks (many strips per block: the weight ring runs across strips and images) -- detections bit for bit"""
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_s_r60"
    imgs = [im.to(dev).half() for im in synth_images(6, 640, 640, seed=77)]
    outs = []
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("YOLORT_AMD_C3T_GEOM", geom)
            monkeypatch.setenv("YOLORT_AMD_C3T_BLOCKS", str(blocks))
        m = YOLOv5(arch=arch, size=(640, 640), score_thresh=0.1)
        m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=0.8))
        m = m.to(dev).half().eval()
        outs.append(m(imgs))
        e = next(iter(m.model._entries.values()))
        assert sum(".tile" in nm for nm in e.plan.names) == 8
    assert sum(len(d["scores"]) for d in outs[0]) > 20
    for a, b in zip(*outs):
        for k in ("boxes", "scores", "labels"):
            assert torch.equal(a[k], b[k]), (geom, k)

