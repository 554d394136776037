"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes), against the
CPU oracle / plain torch fp32 references on the same seeded inputs."""
import ctypes as C

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


def _nhwc(x):  # (n,c,h,w) -> dense NHWC tensor
    return x.permute(0, 2, 3, 1).contiguous()


def _run_conv(dev, dtype, n, cin, cout, h, w, k, s, p, act=True, residual=False, tile=0, x_cs_extra=0, y_cs_extra=0, seed=0):
    from yolort_amd import engine
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=g) * 0.1
    xq, wq = x.to(dtype).float(), wt.to(dtype).float()
    ref = F.conv2d(xq, wq, bias, s, p)
    if act:
        ref = F.silu(ref)
    plan = engine.Plan(dev, dtype)
    # input view with optional extra channel stride (slice of a wider buffer)
    xin = plan.alloc(n, h, w, cin + x_cs_extra, zero=True)
    xv = xin.slice_c(x_cs_extra // 2 if x_cs_extra else 0, cin) if x_cs_extra else xin
    xv.as_tensor().copy_(_nhwc(xq).to(dev, dtype))
    pc = engine.PackedConv(wq, bias, None, dtype, dev)
    ho, wo = engine.conv_out_hw(h, w, (k, k), (s, s), (p, p))
    cpad = (cout + 7) // 8 * 8
    yb = plan.alloc(n, ho, wo, cpad + y_cs_extra, zero=True)
    yv = yb.slice_c(y_cs_extra // 2 if y_cs_extra else 0, cout)
    rv = None
    if residual:
        r = torch.randn(n, cout, ho, wo, generator=g).to(dtype).float()
        ref = ref + r
        rb = plan.alloc(n, ho, wo, cout)
        rb.as_tensor().copy_(_nhwc(r).to(dev, dtype))
        rv = rb
    plan.conv(xv, pc, s, p, act=1 if act else 0, out=yv, res=rv, tile=tile)
    plan.run()
    torch.cuda.synchronize()
    got = yv.as_tensor().float().cpu().permute(0, 3, 1, 2)
    # the reference is the fp32 convolution of the SAME rounded operands, so what remains is the output rounding (2^-11 / 2^-8
    # relative), the fp32 summation order and the hardware exp2 / rcp of SiLU (~1e-6): round 1 allowed 2e-2 here
    tol = 2e-3 if dtype == torch.float16 else (1.6e-2 if dtype == torch.bfloat16 else 4e-6)   # fp32 mode: the summation order and expf are what remain
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * max(1.0, scale), f"max err {err} (scale {scale})"
    if y_cs_extra:  # neighbours in the wider buffer untouched
        full = yb.as_tensor().float().cpu()
        assert full[..., : y_cs_extra // 2].abs().max().item() == 0 and full[..., y_cs_extra // 2 + cout:].abs().max().item() == 0
    return err


def test_mfma_layout_identity(dev):
    """A = I style check with an ASYMMETRIC weight: 1x1 conv whose weight is a permutation-like
    matrix with distinct values catches row/col or k-order swaps in the MFMA fragment mapping."""
    from yolort_amd import engine
    n, c, h, w = 1, 64, 8, 8
    x = torch.arange(n * c * h * w, dtype=torch.float32).reshape(n, c, h, w) % 97 / 97.0
    wt = torch.zeros(64, 64, 1, 1)
    for o in range(64):
        wt[o, (o * 7 + 3) % 64, 0, 0] = 1.0 + o / 64.0
    ref = F.conv2d(x.half().float(), wt.half().float())
    plan = engine.Plan(dev, torch.float16)
    xv = plan.alloc(n, h, w, c)
    xv.as_tensor().copy_(_nhwc(x).to(dev, torch.float16))
    pc = engine.PackedConv(wt, None, None, torch.float16, dev)
    y = plan.conv(xv, pc, 1, 0, act=0)
    plan.run()
    got = y.as_tensor().float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    dict(n=2, cin=64, cout=32, h=40, w=36, k=1, s=1, p=0),
    dict(n=2, cin=32, cout=64, h=33, w=31, k=3, s=1, p=1),
    dict(n=2, cin=32, cout=64, h=40, w=40, k=3, s=2, p=1),
    dict(n=1, cin=128, cout=256, h=20, w=20, k=3, s=2, p=1),
    dict(n=2, cin=16, cout=16, h=24, w=20, k=3, s=1, p=1),      # yolov5n widths
    dict(n=1, cin=48, cout=96, h=16, w=16, k=3, s=2, p=1),      # yolov5m widths (cin % 32 != 0)
    dict(n=3, cin=256, cout=255, h=10, w=10, k=1, s=1, p=0, act=False),  # head
    dict(n=1, cin=512, cout=512, h=20, w=20, k=1, s=1, p=0),
])
def test_conv_parity(dev, dtype, cfg):
    _run_conv(dev, dtype, **cfg)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, -1, -2, -3, -4, -5])
def test_conv_tiles(dev, tile):
    """positive ids: LDS-DMA pipelined kernel (v2) tile configurations; negative: register-staged (v1)"""
    _run_conv(dev, torch.float16, n=2, cin=64, cout=128, h=37, w=29, k=3, s=1, p=1, tile=tile)
    _run_conv(dev, torch.float16, n=2, cin=64, cout=64, h=37, w=29, k=1, s=1, p=0, tile=tile, residual=True)


@pytest.mark.parametrize("tile", [61, 62, 63, 64, 65, 66, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 111, 112, 113, 114, 115, 116, 117, 118, 119])
def test_conv_software_pipelined_tiles(dev, tile):
    """v2 tiles with the software-pipelined main loop (fragment double-buffering, DMA issue between MFMAs):
    short K (1..2 steps, fewer than the ring depth), long K (3x3, 3x3 stride 2), residual, views, ragged M / cout"""
    _run_conv(dev, torch.float16, n=2, cin=32, cout=64, h=23, w=17, k=1, s=1, p=0, tile=tile, seed=tile)               # 1 step
    _run_conv(dev, torch.float16, n=2, cin=64, cout=96, h=37, w=29, k=1, s=1, p=0, tile=tile, residual=True, x_cs_extra=32, y_cs_extra=64, seed=tile + 1)
    _run_conv(dev, torch.float16, n=2, cin=96, cout=128, h=20, w=20, k=1, s=1, p=0, tile=tile, seed=tile + 2)           # 3 steps
    _run_conv(dev, torch.float16, n=2, cin=64, cout=128, h=37, w=29, k=3, s=1, p=1, tile=tile, seed=tile + 3)           # 18 steps
    _run_conv(dev, torch.bfloat16, n=1, cin=128, cout=255, h=21, w=19, k=3, s=2, p=1, tile=tile, act=False, seed=tile + 4)
    _run_conv(dev, torch.float16, n=3, cin=256, cout=64, h=20, w=20, k=1, s=1, p=0, tile=tile, seed=tile + 5)           # 8 steps


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tile", [0, 61, 75, 77, 111, 112, 117, 119, 141, 142, 143, 144, 145, 152])   # 14x / 15x: row-transposed stores, the upsampled copy as whole rows (round 4)
def test_conv_upsampled_second_output(dev, dtype, tile):
    """y2_mode 1: one launch writes the conv output and its nearest x2 upsample (into a channel slice of a wider buffer):
    both must equal the plain conv followed by the upsample kernel, bit for bit"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(41)
    x = torch.randn(3, 96, 13, 11, generator=g).to(dtype).float()
    wt = (torch.randn(64, 96, 1, 1, generator=g) / 10).to(dtype).float()
    b = torch.randn(64, generator=g) * 0.1
    plan = engine.Plan(dev, dtype)
    xv = plan.alloc(3, 13, 11, 96)
    xv.as_tensor().copy_(_nhwc(x).to(dev, dtype))
    pc = engine.PackedConv(wt, b, None, dtype, dev)
    y_ref = plan.conv(xv, pc, 1, 0, tile=tile)
    up_ref = plan.alloc(3, 26, 22, 64)
    plan.upsample2x(y_ref, up_ref)
    y = plan.alloc(3, 13, 11, 64)
    cat = plan.alloc(3, 26, 22, 160, zero=True)
    plan.conv(xv, pc, 1, 0, out=y, up2_out=cat.slice_c(32, 64), tile=tile)
    plan.run()
    assert torch.equal(y.as_tensor(), y_ref.as_tensor())
    got = cat.as_tensor()
    assert torch.equal(got[..., 32:96], up_ref.as_tensor())
    assert got[..., :32].abs().max().item() == 0 and got[..., 96:].abs().max().item() == 0
    ref = torch.nn.functional.silu(torch.nn.functional.conv2d(x, wt, b))
    assert (y.as_tensor().float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < (3e-2 if dtype == torch.bfloat16 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c_,tile", [(32, 0), (32, 63), (32, 76), (32, 13), (32, 114), (32, 121), (64, 0), (64, 62), (64, 72), (64, 12), (64, 80), (64, 81), (64, 113), (64, 122), (128, 0), (128, 78), (128, 79)])
def test_conv_chained_1x1(dev, dtype, c_, tile):
    """chain_w: C3.cv1+cv2 (split output) with the first Bottleneck's 1x1 evaluated in the same launch from the rounded
    outputs in registers -- all three outputs must equal the two-launch form bit for bit"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(51 + c_)
    n, h, w, cin = 2, 19, 23, 64
    x = torch.randn(n, cin, h, w, generator=g).to(dtype).float()
    w1 = (torch.randn(2 * c_, cin, 1, 1, generator=g) / 8).to(dtype).float()
    b1 = torch.randn(2 * c_, generator=g) * 0.1
    w2 = (torch.randn(c_, c_, 1, 1, generator=g) / c_ ** 0.5).to(dtype).float()
    b2 = torch.randn(c_, generator=g) * 0.1
    plan = engine.Plan(dev, dtype)
    xv = plan.alloc(n, h, w, cin)
    xv.as_tensor().copy_(_nhwc(x).to(dev, dtype))
    pc1 = engine.PackedConv(w1, b1, None, dtype, dev)
    pc2 = engine.PackedConv(w2, b2, None, dtype, dev)
    # reference form: two launches
    y_r, cat_r = plan.alloc(n, h, w, c_), plan.alloc(n, h, w, 2 * c_, zero=True)
    plan.conv(xv, pc1, 1, 0, out=y_r, out2=cat_r.slice_c(c_, c_), split=c_)
    t_r = plan.conv(y_r, pc2, 1, 0)
    # chained form
    y, cat, t = plan.alloc(n, h, w, c_), plan.alloc(n, h, w, 2 * c_, zero=True), plan.alloc(n, h, w, c_)
    plan.conv(xv, pc1, 1, 0, out=y, out2=cat.slice_c(c_, c_), split=c_, chain=(pc2, t), tile=tile)
    plan.run()
    torch.cuda.synchronize()
    for name_, got_, ref_ in (("y", y, y_r), ("cat", cat, cat_r), ("t", t, t_r)):
        d_ = (got_.as_tensor().float() - ref_.as_tensor().float()).abs()
        if float(d_.max()) != 0:
            idx_ = torch.nonzero(d_)
            print(f"{name_}: {idx_.shape[0]} of {d_.numel()} elements differ, max {float(d_.max()):.3e}, first {idx_[:6].tolist()}")
    assert torch.equal(y.as_tensor(), y_r.as_tensor())
    assert torch.equal(cat.as_tensor(), cat_r.as_tensor())
    if tile in (121, 122):   # streaming kernel: the chained output may differ from the two-launch form in the last bit of a few elements (measured 6 of 27968)
        d_ = (t.as_tensor().float() - t_r.as_tensor().float()).abs()
        ulp_ = torch.maximum(t_r.as_tensor().float().abs(), torch.tensor(2.0 ** -14, device=d_.device)) * (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7)
        assert bool((d_ <= ulp_).all()) and float((d_ > 0).float().mean()) < 1e-3
    else:
        assert torch.equal(t.as_tensor(), t_r.as_tensor())
    ref = torch.nn.functional.silu(torch.nn.functional.conv2d(torch.nn.functional.silu(torch.nn.functional.conv2d(x, w1[:c_], b1[:c_])).to(dtype).float(), w2, b2))
    assert (t.as_tensor().float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < (6e-2 if dtype == torch.bfloat16 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c_,tile,residual", [(32, 0, True), (32, 33, True), (32, 36, False), (32, 76, True), (64, 0, True), (64, 37, True), (64, 62, False), (64, 81, True)])
def test_conv_chained_over_concat(dev, dtype, c_, tile, residual):
    """C3.cv3 chained to the last Bottleneck.cv2 (3x3 + shortcut): the 1x1 over [fresh outputs | other half of the concat
    buffer] is evaluated in the 3x3 launch's epilogue -- outputs bit-identical to the two-launch form (halo and implicit-GEMM producers)"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(61 + c_)
    n, h, w = 2, 21, 19
    x = torch.randn(n, c_, h, w, generator=g).to(dtype).float()
    w1 = (torch.randn(c_, c_, 3, 3, generator=g) / (3 * c_ ** 0.5)).to(dtype).float()
    b1 = torch.randn(c_, generator=g) * 0.1
    w3 = (torch.randn(2 * c_, 2 * c_, 1, 1, generator=g) / (2 * c_) ** 0.5).to(dtype).float()
    b3 = torch.randn(2 * c_, generator=g) * 0.1
    plan = engine.Plan(dev, dtype)
    xv = plan.alloc(n, h, w, c_)
    xv.as_tensor().copy_(_nhwc(x).to(dev, dtype))
    rv = plan.alloc(n, h, w, c_)
    rv.as_tensor().normal_()
    pc1 = engine.PackedConv(w1, b1, None, dtype, dev)
    pc3 = engine.PackedConv(w3, b3, None, dtype, dev)
    cat_r, cat = plan.alloc(n, h, w, 2 * c_), plan.alloc(n, h, w, 2 * c_)
    cat_r.as_tensor().normal_()
    cat.as_tensor().copy_(cat_r.as_tensor())
    # two launches (same producer tile: halo and implicit-GEMM kernels accumulate K in different orders for cin > 32)
    plan.conv(xv, pc1, 1, 1, out=cat_r.slice_c(0, c_), res=rv if residual else None, tile=tile)
    o_r = plan.conv(cat_r, pc3, 1, 0)
    # one launch
    o = plan.alloc(n, h, w, 2 * c_)
    plan.conv(xv, pc1, 1, 1, out=cat.slice_c(0, c_), res=rv if residual else None, chain=(pc3, o, cat.slice_c(c_, c_)), tile=tile)
    plan.run()
    if tile != 0:
        assert torch.equal(cat.as_tensor(), cat_r.as_tensor())
        assert torch.equal(o.as_tensor(), o_r.as_tensor())
    else:   # autotuned producers may differ between the two forms: fp32 accumulation order only
        assert (cat.as_tensor().float() - cat_r.as_tensor().float()).abs().max().item() < 2e-2
        assert (o.as_tensor().float() - o_r.as_tensor().float()).abs().max().item() < (8e-2 if dtype == torch.bfloat16 else 2e-2)


def test_conv_views_and_residual(dev):
    _run_conv(dev, torch.float16, n=2, cin=64, cout=64, h=20, w=20, k=3, s=1, p=1, residual=True, x_cs_extra=64, y_cs_extra=128)
    _run_conv(dev, torch.float16, n=2, cin=64, cout=32, h=20, w=20, k=1, s=1, p=0, x_cs_extra=32, y_cs_extra=32)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("act_name,residual", [("hardswish", False), ("hardswish", True), ("leaky", False)])
def test_legacy_activation_layer(dev, dtype, act_name, residual):
    """an r3.1 layer = convolution with YMI_ACT_NONE + the activation launch (ymi_act; csrc/preproc_pool.hip): Plan.conv(act=ACT_HARDSWISH / ACT_LEAKY) records both.
    Against torch on the same rounded operands: Hardswish (reference common.py:64-65), LeakyReLU(0.1) (:140), the Bottleneck shortcut added after the activation (:115-116).
    A conv descriptor that names such an activation is refused with a message that points at ymi_act (never a silently linear layer)."""
    import ctypes as C

    from yolort_amd import _lib, engine
    from yolort_amd._lib import ACT_HARDSWISH, ACT_LEAKY
    act, fn = (ACT_HARDSWISH, F.hardswish) if act_name == "hardswish" else (ACT_LEAKY, lambda t: F.leaky_relu(t, 0.1))
    g = torch.Generator().manual_seed(9)
    n, cin, cout, h, w = 2, 64, 96, 20, 24
    x = (torch.randn(n, cin, h, w, generator=g) * 2.0).to(dtype).float()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)).to(dtype).float()
    bias = torch.randn(cout, generator=g) * 0.5
    pre = F.conv2d(x, wt, bias, 1, 1)
    ref = fn(pre.to(dtype).float())          # the pre-activation is stored in the compute dtype between the two launches
    plan = engine.Plan(dev, dtype) if dtype != torch.float32 else engine.Plan(dev, torch.float32)
    xv = plan.alloc(n, h, w, cin, zero=True)
    xv.as_tensor().copy_(_nhwc(x).to(dev, dtype))
    pc = engine.PackedConv(wt, bias, None, dtype, dev)
    cat = plan.alloc(n, h, w, cout + 32, zero=True)
    yv = cat.slice_c(16, cout)               # a channel slice: the neighbours must stay untouched
    rv = None
    if residual:
        r = torch.randn(n, cout, h, w, generator=g).to(dtype).float()
        ref = ref + r
        rv = plan.alloc(n, h, w, cout)
        rv.as_tensor().copy_(_nhwc(r).to(dev, dtype))
    n0 = plan.num_ops
    plan.conv(xv, pc, 1, 1, act=act, out=yv, res=rv)
    assert plan.num_ops == n0 + 2 and plan.meta[-1]["kind"] == "act" and plan.io[n0]["post_act"]["act"] == act
    plan.run()
    torch.cuda.synchronize()
    got = yv.as_tensor().float().cpu().permute(0, 3, 1, 2)
    tol = 2e-3 if dtype == torch.float16 else (1.6e-2 if dtype == torch.bfloat16 else 4e-6)
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    full = cat.as_tensor().float().cpu()
    assert full[..., :16].abs().max().item() == 0 and full[..., 16 + cout:].abs().max().item() == 0
    d = plan.conv_descs[n0]
    d.act = act
    lib = _lib.load(require_gpu=True)
    assert lib.ymi_conv2d(C.byref(d), _lib.stream_ptr()) != 0 and b"ymi_act" in lib.ymi_last_error()


def test_conv_second_output(dev):
    """one launch, two destinations: couts [0,32) -> y, [32,64) -> channel slice of another buffer (C3 cv1+cv2)"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 64, 24, 20, generator=g).half().float()
    wt = (torch.randn(64, 64, 1, 1, generator=g) / 8).half().float()
    b = torch.randn(64, generator=g) * 0.1
    ref = F.silu(F.conv2d(x, wt, b))
    plan = engine.Plan(dev, torch.float16)
    xv = plan.alloc(2, 24, 20, 64)
    xv.as_tensor().copy_(_nhwc(x).to(dev, torch.float16))
    y = plan.alloc(2, 24, 20, 32)
    cat = plan.alloc(2, 24, 20, 64, zero=True)
    plan.conv(xv, engine.PackedConv(wt, b, None, torch.float16, dev), 1, 0, out=y, out2=cat.slice_c(32, 32), split=32)
    plan.run()
    got1 = y.as_tensor().float().cpu().permute(0, 3, 1, 2)
    got2 = cat.as_tensor().float().cpu().permute(0, 3, 1, 2)
    assert (got1 - ref[:, :32]).abs().max().item() < 2e-2
    assert (got2[:, 32:] - ref[:, 32:]).abs().max().item() < 2e-2
    assert got2[:, :32].abs().max().item() == 0


@pytest.mark.parametrize("k,s,p,cin,cout,hw", [(1, 1, 0, 64, 32, 160), (3, 1, 1, 32, 32, 96), (3, 2, 1, 32, 64, 128), (3, 1, 1, 128, 128, 40), (1, 1, 0, 512, 256, 20)])
def test_conv_v1_v2_agree_at_scale(dev, k, s, p, cin, cout, hw):
    """larger, multi-block problems: pipelined (v2) and register-staged (v1) kernels against torch fp32"""
    for tile in (0, -100):
        _run_conv(dev, torch.float16, n=2, cin=cin, cout=cout, h=hw, w=hw, k=k, s=s, p=p, tile=tile, seed=k + cin)


@pytest.mark.parametrize("stride", [1, 2])
def test_row_streaming_3x3_counted_wait_equals_the_full_drain_over_many_launches(dev, stride, monkeypatch):
    """conv3x3_rs.hip (tiles 137 / 138, both in the pinned yolov5s plan): the counted `s_waitcnt vmcnt(N)` that keeps D - 1 row groups in flight across the step barrier
    against YOLORT_AMD_RS_COUNTED=0 (every step drains the counter): 200 back-to-back launches on a batch-32 layer while a second stream saturates the memory system with
    copies -- every launch bit-identical to the drained form (the round-4 count assumed stores never overtake older loads; the count now relies on load
    order only)"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(137 + stride)
    n, cin, h, w = 32, 64, 80 if stride == 1 else 160, 80 if stride == 1 else 160
    cout = 64 if stride == 1 else 128
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)).half()
    bias = torch.randn(cout, generator=g) * 0.1
    plan = engine.Plan(dev, torch.float16)
    xv = plan.alloc(n, h, w, cin)
    xv.as_tensor().copy_(_nhwc(x.float()).to(dev, torch.float16))
    pc = engine.PackedConv(wt.float(), bias, None, torch.float16, dev)
    y = plan.conv(xv, pc, stride, 1, tile=136 + stride)
    monkeypatch.setenv("YOLORT_AMD_RS_COUNTED", "0")
    plan.run()
    torch.cuda.synchronize()
    ref = y.as_tensor().clone()
    assert float(ref.float().abs().max()) > 0
    monkeypatch.delenv("YOLORT_AMD_RS_COUNTED")
    noise_a, noise_b = torch.empty(64 << 20, device=dev, dtype=torch.uint8), torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    side = torch.cuda.Stream(device=dev)
    bad = 0
    for it in range(200):
        with torch.cuda.stream(side):
            noise_b.copy_(noise_a, non_blocking=True)      # memory traffic from another queue while the kernel's row groups are in flight
        y.as_tensor().zero_()
        plan.run()
        torch.cuda.synchronize()
        bad += int(not torch.equal(y.as_tensor(), ref))
    assert bad == 0, f"{bad} of 200 launches differ from the fully drained form"


@pytest.mark.parametrize("tile", [201, 202, 203, 204, 205, 206, 0, -100])
@pytest.mark.parametrize("shape", [dict(n=2, cin=64, cout=32, h=160, w=160, k=1, s=1, p=0), dict(n=2, cin=32, cout=32, h=96, w=96, k=3, s=1, p=1), dict(n=2, cin=32, cout=64, h=128, w=128, k=3, s=2, p=1),
                                   dict(n=2, cin=128, cout=128, h=40, w=40, k=3, s=1, p=1), dict(n=2, cin=512, cout=256, h=20, w=20, k=1, s=1, p=0), dict(n=1, cin=48, cout=96, h=33, w=21, k=3, s=2, p=1),
                                   dict(n=3, cin=24, cout=40, h=17, w=35, k=3, s=1, p=1)])
def test_conv_f32_pipelined_tiles(dev, tile, shape):
    """fp32 mode (csrc/conv_f32_pipe.hip, round 5): every LDS-DMA pipelined fp32 tile, the library's shape rule (0) and the register-staged kernel it replaces (-100) against
    torch's fp32 convolution to rounding-order accuracy -- pointwise, uniform-tap and im2col-table operand forms, ragged sizes, shortcut, channel-slice views"""
    _run_conv(dev, torch.float32, tile=tile, residual=True, x_cs_extra=32, y_cs_extra=64, seed=tile % 7 + shape["cin"], **shape)
    _run_conv(dev, torch.float32, tile=tile, seed=tile % 5 + shape["h"], **shape)


@pytest.mark.parametrize("shape", [dict(n=2, cin=64, cout=64, h=80, w=80, k=3, s=1, p=1), dict(n=2, cin=256, cout=512, h=40, w=40, k=3, s=2, p=1), dict(n=2, cin=1024, cout=512, h=20, w=20, k=1, s=1, p=0),
                                   dict(n=1, cin=8, cout=32, h=64, w=48, k=3, s=1, p=1), dict(n=2, cin=768, cout=1024, h=20, w=20, k=3, s=2, p=1)])
def test_conv_f32_pipelined_tiles_equal_the_register_staged_kernel_bit_for_bit(dev, shape):
    """every pipelined fp32 tile feeds the f32-input MFMA the same k pairs in the same order and folds the same 64-element partial sums as csrc/conv_f32.hip: the
    outputs are equal to the last bit -- the detections validated against the reference-made goldens in rounds 3-4 do not move with the kernel family or the tile"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(shape["cin"] + shape["h"])
    n, cin, cout, h, w, k, s_, p = (shape[q] for q in ("n", "cin", "cout", "h", "w", "k", "s", "p"))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=g) * 0.1
    plan = engine.Plan(dev, torch.float32)
    xv = plan.alloc(n, h, w, cin)
    xv.as_tensor().copy_(_nhwc(x).to(dev))
    pc = engine.PackedConv(wt, bias, None, torch.float32, dev)
    outs = {t: plan.conv(xv, pc, s_, p, tile=t) for t in (-100, 0, 201, 202, 203, 204, 205, 206)}
    plan.run()
    torch.cuda.synchronize()
    ref = outs[-100].as_tensor()
    assert float(ref.abs().max()) > 0
    for t, v in outs.items():
        assert torch.equal(v.as_tensor(), ref), f"tile {t}: max difference {(v.as_tensor() - ref).abs().max().item()}"


def test_conv_f32_split_and_upsampled_outputs(dev):
    """fp32 mode: C3.cv1 + cv2 as one launch (channel split into a slice of the concat buffer, common.py:172-173) and the PAN's 1x1 Conv with nn.Upsample folded into its
    epilogue (path_aggregation_network.py:221-223): equal to the separate fp32 launches bit for bit (same tile, same summation order)"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(5)
    n, cin, h, w = 2, 128, 40, 40
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(128, cin, 1, 1, generator=g) / np.sqrt(cin)
    bias = torch.randn(128, generator=g) * 0.1
    plan = engine.Plan(dev, torch.float32)
    xv = plan.alloc(n, h, w, cin)
    xv.as_tensor().copy_(_nhwc(x).to(dev))
    tile = 202
    pair = engine.PackedConv(wt, bias, None, torch.float32, dev)
    cat = plan.alloc(n, h, w, 128, zero=True)
    y1 = plan.alloc(n, h, w, 64)
    plan.conv(xv, pair, 1, 0, out=y1, out2=cat.slice_c(64, 64), split=64, tile=tile, name="pair")
    a = plan.conv(xv, engine.PackedConv(wt[:64], bias[:64], None, torch.float32, dev), 1, 0, tile=tile)
    b = plan.conv(xv, engine.PackedConv(wt[64:], bias[64:], None, torch.float32, dev), 1, 0, tile=tile)
    up = plan.alloc(n, 2 * h, 2 * w, 192, zero=True)
    y2 = plan.conv(xv, pair, 1, 0, up2_out=up.slice_c(0, 128), tile=tile, name="up")
    plan.run()
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(x, wt, bias)).permute(0, 2, 3, 1)
    assert (y2.as_tensor().cpu() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    assert torch.equal(y1.as_tensor(), a.as_tensor()) and torch.equal(cat.as_tensor()[..., 64:], b.as_tensor()) and float(cat.as_tensor()[..., :64].abs().max()) == 0.0
    assert torch.equal(torch.cat([a.as_tensor(), b.as_tensor()], -1), y2.as_tensor())
    u = up.as_tensor()
    assert torch.equal(u[..., :128], y2.as_tensor().repeat_interleave(2, 1).repeat_interleave(2, 2)) and float(u[..., 128:].abs().max()) == 0.0


@pytest.mark.parametrize("variant", [31, 32, 33, 34, 35, 36, 37])
@pytest.mark.parametrize("shape", [dict(n=2, cin=32, cout=32, h=40, w=40), dict(n=1, cin=64, cout=64, h=33, w=21), dict(n=2, cin=128, cout=128, h=20, w=20),
                                   dict(n=1, cin=32, cout=64, h=8, w=16), dict(n=3, cin=64, cout=96, h=17, w=35)])
def test_conv3x3_halo_kernel(dev, variant, shape):
    """LDS-halo 3x3 s1 kernel (activation patch resident in LDS across the nine taps), incl. ragged
    image sizes (partial patches), residual and channel-slice views"""
    _run_conv(dev, torch.float16, k=3, s=1, p=1, tile=variant, residual=True, x_cs_extra=32, y_cs_extra=64, seed=variant, **shape)
    _run_conv(dev, torch.bfloat16, k=3, s=1, p=1, tile=variant, seed=variant + 1, **shape)


@pytest.mark.parametrize("variant", [91, 92, 93, 94, 95])
@pytest.mark.parametrize("shape", [dict(n=2, cin=32, cout=32, h=40, w=40), dict(n=1, cin=64, cout=64, h=33, w=21), dict(n=2, cin=128, cout=128, h=20, w=20),
                                   dict(n=1, cin=32, cout=64, h=8, w=16), dict(n=3, cin=64, cout=96, h=17, w=35), dict(n=1, cin=64, cout=128, h=80, w=80),
                                   dict(n=2, cin=32, cout=32, h=5, w=7), dict(n=1, cin=96, cout=160, h=24, w=52)])
def test_conv_halo8_kernel(dev, variant, shape):
    """8-wave LDS-halo 3x3 s1 kernel (conv_halo8.hip): every wave layout / ring depth, patch shapes chosen per map size
    (16x16, 10x20, 6x40, whole tiny maps), ragged sizes, several cout blocks, residual and channel-slice views"""
    _run_conv(dev, torch.float16, k=3, s=1, p=1, tile=variant, residual=True, x_cs_extra=32, y_cs_extra=64, seed=variant, **shape)
    _run_conv(dev, torch.bfloat16, k=3, s=1, p=1, tile=variant, seed=variant + 1, **shape)


def test_conv_igemm8_192_cout_blocks(dev):
    """tile 120 (conv_igemm8_kernel<.., 192, 4, ..>): yolov5m's 96 -> 192 and 192 -> 192 stride-2 3x3 shapes at a reduced batch and a 1x1 with a shortcut; against torch and equal to
    tile 111 bit for bit"""
    from yolort_amd import engine
    for dtype, (n, cin, cout, h, w, k, s_, res) in [(torch.bfloat16, (2, 96, 192, 160, 160, 3, 2, False)), (torch.float16, (2, 192, 192, 80, 80, 3, 2, False)), (torch.bfloat16, (1, 192, 384, 37, 29, 1, 1, True))]:
        g = torch.Generator().manual_seed(120 + cin + h)
        p_ = k // 2
        x = torch.randn(n, cin, h, w, generator=g).to(dtype).float()
        wt = (torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)).to(dtype).float()
        bias = torch.randn(cout, generator=g) * 0.1
        ref = F.silu(F.conv2d(x, wt, bias, s_, p_))
        r = torch.randn(*ref.shape, generator=g).to(dtype).float() if res else None
        if res:
            ref = ref + r
        outs = []
        for tile in (111, 120):
            plan = engine.Plan(dev, dtype)
            xv = plan.alloc(n, h, w, cin)
            xv.as_tensor().copy_(_nhwc(x).to(dev, dtype))
            pc = engine.PackedConv(wt, bias, None, dtype, dev)
            ho, wo = engine.conv_out_hw(h, w, (k, k), (s_, s_), (p_, p_))
            yv = plan.alloc(n, ho, wo, cout, zero=True)
            rv = None
            if res:
                rv = plan.alloc(n, ho, wo, cout)
                rv.as_tensor().copy_(_nhwc(r).to(dev, dtype))
            plan.conv(xv, pc, s_, p_, out=yv, res=rv, tile=tile)
            plan.run()
            torch.cuda.synchronize()
            outs.append(yv.as_tensor().clone())
        got = outs[1].float().cpu().permute(0, 3, 1, 2)
        tol = 2e-3 if dtype == torch.float16 else 1.6e-2
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_conv_halo8_96_cout_blocks(dev):
    """tile 96 (conv_halo8_kernel<.., 96, 8>): yolov5m's 96 -> 96 / 192 -> 192 3x3 shapes at a reduced batch, with a shortcut; against torch and equal to tile 91 bit for bit"""
    from yolort_amd import engine
    for dtype, (n, cin, cout, h, w, res) in [(torch.bfloat16, (2, 96, 96, 160, 160, True)), (torch.float16, (2, 192, 192, 80, 80, False)), (torch.bfloat16, (1, 96, 192, 37, 29, True))]:
        g = torch.Generator().manual_seed(96 + cin + h)
        x = torch.randn(n, cin, h, w, generator=g).to(dtype).float()
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)).to(dtype).float()
        bias = torch.randn(cout, generator=g) * 0.1
        ref = F.silu(F.conv2d(x, wt, bias, 1, 1))
        r = torch.randn(n, cout, h, w, generator=g).to(dtype).float() if res else None
        if res:
            ref = ref + r
        outs = []
        for tile in (91, 96):
            plan = engine.Plan(dev, dtype)
            xv = plan.alloc(n, h, w, cin)
            xv.as_tensor().copy_(_nhwc(x).to(dev, dtype))
            pc = engine.PackedConv(wt, bias, None, dtype, dev)
            yv = plan.alloc(n, h, w, cout, zero=True)
            rv = None
            if res:
                rv = plan.alloc(n, h, w, cout)
                rv.as_tensor().copy_(_nhwc(r).to(dev, dtype))
            plan.conv(xv, pc, 1, 1, out=yv, res=rv, tile=tile)
            plan.run()
            torch.cuda.synchronize()
            outs.append(yv.as_tensor().clone())
        got = outs[1].float().cpu().permute(0, 3, 1, 2)
        tol = 2e-3 if dtype == torch.float16 else 1.6e-2
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("cout", [32, 64])
@pytest.mark.parametrize("shape", [dict(n=2, h=40, w=40), dict(n=1, h=33, w=21), dict(n=3, h=8, w=16), dict(n=2, h=5, w=7), dict(n=9, h=64, w=96), dict(n=1, h=160, w=160)])
def test_conv3x3_c32_kernel(dev, stride, cout, shape):
    """resident-weights persistent 3x3 kernel for cin = 32 (conv3x3_c32.hip, tile 131): stride 1 / 2 (parity-split patch columns),
    cout 32 / 64, ragged sizes (partial tiles, odd widths), more tiles than resident blocks (the persistent loop), residual
    (stride 1) and channel-slice views"""
    _run_conv(dev, torch.float16, cin=32, cout=cout, k=3, s=stride, p=1, tile=131, residual=(stride == 1 and cout == 32), x_cs_extra=32, y_cs_extra=64,
              seed=131 + stride + cout, **shape)
    _run_conv(dev, torch.bfloat16, cin=32, cout=cout, k=3, s=stride, p=1, tile=131, seed=132 + stride + cout, **shape)


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 48), (48, 48), (64, 32), (48, 64)])
@pytest.mark.parametrize("shape", [dict(n=2, h=40, w=40), dict(n=1, h=33, w=21), dict(n=3, h=16, w=16), dict(n=2, h=5, w=7), dict(n=12, h=80, w=80), dict(n=2, h=160, w=160)])
def test_conv3x3_res_kernel(dev, cin, cout, shape):
    """resident-weights persistent 3x3 kernel for cin = 48 / 64 (conv3x3_res.hip, tile 132): ragged sizes (partial 16 x 16 tiles), more tiles than
    resident blocks (12 x 25 = 300 tiles on 256 blocks: the persistent loop, its double-buffered patch and deferred stores), cout below the 64-wide wave tile, residual and channel-slice views"""
    for tile in (132, 133) if cout == 64 else (132,):   # 133 = the register-weights variant (conv3x3_rw.hip): cout = 64 only
        _run_conv(dev, torch.float16, cin=cin, cout=cout, k=3, s=1, p=1, tile=tile, residual=True, x_cs_extra=32 if cin == 64 else 16, y_cs_extra=64, seed=132 + cin + cout, **shape)
        _run_conv(dev, torch.bfloat16, cin=cin, cout=cout, k=3, s=1, p=1, tile=tile, seed=133 + cin + cout, **shape)


@pytest.mark.parametrize("shape", [dict(n=2, h=40, w=40), dict(n=1, h=33, w=21), dict(n=3, h=16, w=16), dict(n=2, h=5, w=7), dict(n=12, h=160, w=160), dict(n=2, h=161, w=320)])
def test_conv3x3_rw2_kernel(dev, shape):
    """stride-2 register-weights 3x3 kernel, 64 -> 128 (conv3x3_rw2.hip, tile 134): ragged sizes against the 8 x 8 tiles (odd inputs), more tiles than resident
    blocks (12 x 100 = 1200 tiles on 512 blocks: the persistent loop and its double-buffered parity-split patch), channel-slice views on both sides"""
    for tile in (134,):
        _run_conv(dev, torch.float16, cin=64, cout=128, k=3, s=2, p=1, tile=tile, x_cs_extra=32, y_cs_extra=64, seed=134, **shape)
        _run_conv(dev, torch.bfloat16, cin=64, cout=128, k=3, s=2, p=1, tile=tile, seed=135, **shape)


@pytest.mark.parametrize("cout", [128, 256])
@pytest.mark.parametrize("shape", [dict(n=2, h=40, w=40), dict(n=1, h=33, w=21), dict(n=2, h=5, w=7), dict(n=12, h=80, w=80), dict(n=3, h=81, w=160)])
def test_conv3x3_rw3_ksplit_kernel(dev, cout, shape):
    """tile 135 (conv3x3_rw2.hip, round 4): 128 -> 128 / 256 at stride 2, K split over two waves (partial sums through the consumed patch buffer), one 8-wave block
    per CU walking the tiles; ragged maps, more tiles than resident blocks (12 x 25 = 300 tiles on 256 / 128 blocks), channel-slice views on both sides -- against the
    fp32 convolution of the same rounded operands"""
    _run_conv(dev, torch.float16, cin=128, cout=cout, k=3, s=2, p=1, tile=135, x_cs_extra=32, y_cs_extra=64, seed=135 + cout, **shape)
    _run_conv(dev, torch.bfloat16, cin=128, cout=cout, k=3, s=2, p=1, tile=135, seed=136 + cout, **shape)


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("shape", [dict(n=2, h=40, w=40), dict(n=1, h=33, w=21), dict(n=3, h=16, w=16), dict(n=2, h=5, w=7), dict(n=12, h=80, w=80), dict(n=4, h=161, w=160)])
def test_conv3x3_rs_kernel(dev, stride, shape, monkeypatch):
    """row-streaming 3x3 (conv3x3_rs.hip, tiles 137 / 138): strips x chunks of steps over the concatenated images, ring of rows, counted waits (and the drain-everything form:
    YOLORT_AMD_RS_COUNTED is read once per process, so the counted form is what runs here); ragged maps, channel-slice views; bit-identical to the implicit GEMM"""
    from yolort_amd import engine
    cout, tile, ref_tile = (64, 137, 113) if stride == 1 else (128, 138, 111)
    _run_conv(dev, torch.float16, cin=64, cout=cout, k=3, s=stride, p=1, tile=tile, x_cs_extra=32, y_cs_extra=64, seed=137 + stride, **shape)
    _run_conv(dev, torch.bfloat16, cin=64, cout=cout, k=3, s=stride, p=1, tile=tile, seed=138 + stride, **shape)
    n, h, w = shape["n"], shape["h"], shape["w"]
    g = torch.Generator().manual_seed(140 + h)
    x = torch.randn(n, 64, h, w, generator=g)
    wt = torch.randn(cout, 64, 3, 3, generator=g) / np.sqrt(64 * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    outs = []
    for t in (tile, ref_tile):
        plan = engine.Plan(dev, torch.float16)
        xv = plan.alloc(n, h, w, 64)
        xv.as_tensor().copy_(_nhwc(x).to(dev, torch.float16))
        pc = engine.PackedConv(wt.half().float(), bias, None, torch.float16, dev)
        ho, wo = engine.conv_out_hw(h, w, (3, 3), (stride, stride), (1, 1))
        yv = plan.alloc(n, ho, wo, cout, zero=True)
        plan.conv(xv, pc, stride, 1, out=yv, tile=t)
        for _ in range(3):   # (repeated: a missed wait shows as a run-to-run difference)
            plan.run()
        torch.cuda.synchronize()
        outs.append(yv.as_tensor().clone())
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_conv3x3_rw2_equals_the_implicit_gemm_bit_for_bit(dev):
    """tile 134 accumulates in the implicit GEMM's K order and rounds through the same lean epilogue: equal to tiles 111 / 143 bit for bit (yolov5s body.3's shape
    at a reduced batch and a ragged variant of it)"""
    from yolort_amd import engine
    for (n, h, w) in ((4, 160, 160), (3, 77, 91)):
        g = torch.Generator().manual_seed(134 + h)
        x = torch.randn(n, 64, h, w, generator=g)
        wt = torch.randn(128, 64, 3, 3, generator=g) / np.sqrt(64 * 9)
        bias = torch.randn(128, generator=g) * 0.1
        outs = []
        for tile in (134, 111, 143):
            plan = engine.Plan(dev, torch.float16)
            xv = plan.alloc(n, h, w, 64)
            xv.as_tensor().copy_(_nhwc(x).to(dev, torch.float16))
            pc = engine.PackedConv(wt.half().float(), bias, None, torch.float16, dev)
            ho, wo = engine.conv_out_hw(h, w, (3, 3), (2, 2), (1, 1))
            yv = plan.alloc(n, ho, wo, 128, zero=True)
            plan.conv(xv, pc, 2, 1, out=yv, tile=tile)
            plan.run()
            torch.cuda.synchronize()
            outs.append(yv.as_tensor().clone())
        for o in outs[1:]:
            assert torch.equal(outs[0].view(torch.int16), o.view(torch.int16))


@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("cout", [64, 32])
def test_conv3x3_res_equals_the_implicit_gemm_bit_for_bit(dev, cout, residual):
    """tile 132 accumulates in the implicit GEMM's K order (tap-major, channel-minor) and rounds through the same silu_pair / lane-swap code, its epilogue
    spread over the next tile's MFMA loop and its shortcut fetched in packet form: the outputs equal tile 113 / 114's bit for bit (300 tiles on 256 blocks:
    deferred epilogues, the tail, partial tiles)"""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(7 + cout)
    n, h, w, cin = 12, 80, 77, 64
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(n, cout, h, w, generator=g)
    outs = []
    for tile in (132, 113 if cout > 32 else 114) + ((133,) if cout == 64 else ()):   # 133: the register-weights variant (conv3x3_rw.hip, opt-in)
        plan = engine.Plan(dev, torch.float16)
        xv = plan.alloc(n, h, w, cin)
        xv.as_tensor().copy_(_nhwc(x).to(dev, torch.float16))
        pc = engine.PackedConv(wt.half().float(), bias, None, torch.float16, dev)
        yv = plan.alloc(n, h, w, cout, zero=True)
        rv = None
        if residual:
            rv = plan.alloc(n, h, w, cout)
            rv.as_tensor().copy_(_nhwc(r).to(dev, torch.float16))
        plan.conv(xv, pc, 1, 1, out=yv, res=rv, tile=tile)
        plan.run()
        torch.cuda.synchronize()
        outs.append(yv.as_tensor().clone())
    for o in outs[1:]:
        assert torch.equal(outs[0].view(torch.int16), o.view(torch.int16)), f"max difference {(outs[0].float() - o.float()).abs().max().item()}"


@pytest.mark.parametrize("tile", [121, 122, 123, 124])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv1x1_stream_kernel(dev, dtype, tile):
    """streaming 1x1 kernel (weights in registers, activations global -> VGPR, no LDS): every cout-block width that has an
    instance for the K at hand, ragged pixel counts (clamped lanes), channel-slice views, residual, several groups per wave"""
    tnw = tile - 120
    for cin, cout, h, w in [(64, 64, 37, 29), (32, 96, 23, 17), (128, 64, 20, 20), (96, 96, 31, 9), (64, 128, 160, 160), (128, 160, 9, 9)]:
        n = 3 if h < 100 else 8
        _run_conv(dev, dtype, n=n, cin=cin, cout=cout, h=h, w=w, k=1, s=1, p=0, tile=tile, residual=(cin == 64), x_cs_extra=32, y_cs_extra=64, seed=tile + cin)


def test_conv_head_fp32_out(dev):
    from yolort_amd import engine
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 128, 12, 12, generator=g).half().float()
    wt = (torch.randn(255, 128, 1, 1, generator=g) / 11).half().float()
    b = torch.randn(255, generator=g)
    ref = F.conv2d(x, wt, b)
    plan = engine.Plan(dev, torch.float16)
    xv = plan.alloc(2, 12, 12, 128)
    xv.as_tensor().copy_(_nhwc(x).to(dev, torch.float16))
    y = plan.conv(xv, engine.PackedConv(wt, b, None, torch.float16, dev), 1, 0, act=0, out_dtype=torch.float32)
    assert y.dtype == torch.float32 and y.cs == 256 and y.c == 255
    plan.run()
    got = y.as_tensor().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("tile", [0, 13, 23, 26, 41, -100])
@pytest.mark.parametrize("cout,hw", [(32, (64, 96)), (48, (96, 160)), (16, (32, 64)), (64, (160, 224))])
def test_stem_superpixel(dev, tile, cout, hw):
    """Conv(3, c, k=6, s=2, p=2) (darknetv6.py:81) through the NHWC4 super-pixel formulation: generic
    implicit-GEMM tiles, the dedicated stem kernel (41) and the autotuned choice (0)."""
    from yolort_amd import engine
    g = torch.Generator().manual_seed(5 + cout)
    x = torch.rand(2, 3, hw[0], hw[1], generator=g).half().float()
    wt = (torch.randn(cout, 3, 6, 6, generator=g) / 10).half().float()
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.silu(F.conv2d(x, wt, b, 2, 2))
    plan = engine.Plan(dev, torch.float16)
    xv = plan.alloc(2, hw[0], hw[1], 4, zero=True)
    xv.as_tensor()[..., :3].copy_(_nhwc(x).to(dev, torch.float16))
    pc = engine.PackedConv(wt, b, None, torch.float16, dev, stem_superpixel=True)
    if tile in (13, 23, 26) and cout > 32:
        pytest.skip("32-wide cout tiles only")
    y = plan.conv(xv, pc, 2, 2, tile=tile)
    plan.run()
    got = y.as_tensor().float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-2


def test_bn_fold(dev):
    from yolort_amd import engine
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 32, 16, 16, generator=g).half().float()
    wt = (torch.randn(64, 32, 3, 3, generator=g) / 17)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    mean, var = torch.randn(64, generator=g) * 0.3, torch.rand(64, generator=g) + 0.2
    ref = F.silu(F.batch_norm(F.conv2d(x, wt, None, 1, 1), mean, var, gamma, beta, False, 0.0, 1e-3))
    plan = engine.Plan(dev, torch.float16)
    xv = plan.alloc(1, 16, 16, 32)
    xv.as_tensor().copy_(_nhwc(x).to(dev, torch.float16))
    y = plan.conv(xv, engine.PackedConv(wt, None, (gamma, beta, mean, var, 1e-3), torch.float16, dev), 1, 1)
    plan.run()
    got = y.as_tensor().float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("spp_g", ["default", "1", "4"])
def test_spp_pool_and_upsample_exact(dev, spp_g, monkeypatch):
    from yolort_amd import engine
    monkeypatch.delenv("YOLORT_AMD_SPP_G", raising=False)
    if spp_g != "default":
        monkeypatch.setenv("YOLORT_AMD_SPP_G", spp_g)   # channel groups per block of the LDS cascade kernel
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 20, 17, generator=g).half()
    plan = engine.Plan(dev, torch.float16)
    buf = plan.alloc(2, 20, 17, 256, zero=True)
    buf.slice_c(0, 64).as_tensor().copy_(_nhwc(x.float()).to(dev, torch.float16))
    plan.spp_pool(buf, 64)
    up = plan.alloc(2, 40, 34, 96, zero=True)
    plan.upsample2x(buf.slice_c(0, 64), up.slice_c(32, 64))
    plan.run()
    got = buf.as_tensor().float().cpu().permute(0, 3, 1, 2)
    xf = x.float()
    for i, k in enumerate((5, 9, 13)):
        ref = F.max_pool2d(xf, k, 1, k // 2)
        assert torch.equal(got[:, 64 * (i + 1): 64 * (i + 2)], ref), f"maxpool{k} not exact"
    gu = up.as_tensor().float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(gu[:, 32:96], F.interpolate(xf, scale_factor=2.0, mode="nearest"))
    assert gu[:, :32].abs().max().item() == 0
    # bfloat16 (round 4: order keys + packed unsigned maxima): negative values, zeros of both signs, magnitudes over sixty orders, a 40 x 40 map (yolov5m's shape)
    for (n, c, h, w) in ((2, 32, 13, 21), (1, 96, 40, 40)):
        xb = torch.randn(n, c, h, w, generator=g) * torch.tensor([1e-30, 1.0, 3e4, 1e30]).repeat(c // 4).view(1, c, 1, 1)
        xb[0, :, 3:6, 4:9] = 0.0
        xb[0, :, 4, 5] = -0.0
        xb[-1, 5] = -xb[-1, 5].abs() - 1.0
        xb = xb.to(torch.bfloat16)
        pb = engine.Plan(dev, torch.bfloat16)
        bb = pb.alloc(n, h, w, 4 * c, zero=True)
        bb.slice_c(0, c).as_tensor().copy_(_nhwc(xb).to(dev))
        pb.spp_pool(bb, c)
        pb.run()
        gb = bb.as_tensor().float().cpu().permute(0, 3, 1, 2)
        assert torch.equal(gb[:, :c], xb.float())
        for i, k in enumerate((5, 9, 13)):
            assert torch.equal(gb[:, c * (i + 1): c * (i + 2)], F.max_pool2d(xb.float(), k, 1, k // 2)), f"bf16 maxpool{k} not exact ({h}x{w})"
    # fp32 mode (round 5: LDS cascade; the 80 x 80 plane does not fit twice in LDS and takes the direct kernel)
    for (n, c, h, w) in ((2, 64, 20, 20), (1, 96, 40, 40), (1, 8, 80, 80)):
        xf32 = torch.randn(n, c, h, w, generator=g)
        xf32[-1, 5] = -xf32[-1, 5].abs() - 1.0
        pf = engine.Plan(dev, torch.float32)
        bf = pf.alloc(n, h, w, 4 * c, zero=True)
        bf.slice_c(0, c).as_tensor().copy_(_nhwc(xf32).to(dev))
        pf.spp_pool(bf, c)
        pf.run()
        gf = bf.as_tensor().cpu().permute(0, 3, 1, 2)
        assert torch.equal(gf[:, :c], xf32)
        for i, k in enumerate((5, 9, 13)):
            assert torch.equal(gf[:, c * (i + 1): c * (i + 2)], F.max_pool2d(xf32, k, 1, k // 2)), f"fp32 maxpool{k} not exact ({h}x{w})"


def test_letterbox_vs_oracle(dev):
    from oracle import yolov5_oracle as O
    from yolort_amd.models.transform import YOLOTransform
    from workloads.synth import synth_images
    shapes = [(1080, 810), (480, 640), (720, 1280), (375, 500), (100, 37), (641, 480)]
    imgs = [synth_images(1, h, w, seed=h + w)[0] for h, w in shapes]
    ref, sizes = O.letterbox(imgs, 640, 640, 32)
    t = YOLOTransform(640, 640)
    nt, _ = t([im.to(dev) for im in imgs], None, dtype=torch.float32)
    assert nt.image_sizes == sizes
    got = nt.nchw().float().cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 5e-5
    # fp16 output path + uint8 input path
    nt16, _ = t([im.to(dev) for im in imgs], None, dtype=torch.float16)
    assert (nt16.nchw().float().cpu() - ref).abs().max().item() <= 1e-3
    u8 = [(im * 255).round().to(torch.uint8) for im in imgs]
    ref8, _ = O.letterbox([u.float() / 255.0 for u in u8], 640, 640, 32)
    nt8, _ = t([u.to(dev) for u in u8], None, dtype=torch.float32)
    assert (nt8.nchw().float().cpu() - ref8).abs().max().item() <= 5e-5


@pytest.mark.parametrize("kernel", ["default", "1", "2", "4", "2+cg2", "4+cg2", "1+cg2"])
@pytest.mark.parametrize("in_dtype,hwc", [(torch.float16, False), (torch.bfloat16, False), (torch.float32, False), (torch.uint8, False), (torch.uint8, True)])
@pytest.mark.parametrize("out_dtype", [torch.float16, torch.bfloat16])   # (bf16: the tiled kernels round with v_cvt_pk_bf16_f32, the per-pixel kernel in software)
def test_letterbox_tiled_kernel_equals_per_pixel_kernel(dev, in_dtype, hwc, kernel, out_dtype, monkeypatch):
    """round 2: the tiled, LDS-staged letterbox kernels (16-byte source loads, two pixels per store; the lean one with 1 / 2 / 4
    rows per wave and the first tiled kernel) must reproduce the per-pixel kernel bit for bit -- up / down scaling, odd sizes,
    rows that are not 16-byte aligned, every input type; the 8-channel output form still takes the per-pixel kernel and serves as
    the reference"""
    monkeypatch.delenv("YOLORT_AMD_LETTERBOX", raising=False)
    monkeypatch.delenv("YOLORT_AMD_LB_CG", raising=False)
    monkeypatch.delenv("YOLORT_AMD_LB_BLOCKS", raising=False)
    if kernel != "default":
        monkeypatch.setenv("YOLORT_AMD_LETTERBOX", kernel.split("+")[0])
        monkeypatch.setenv("YOLORT_AMD_LB_CG", "2" if kernel.endswith("cg2") else "1")
        if "+b" in kernel:   # the persistent DMA-staged kernel with a handful of blocks: each walks many tiles of several images through both staging buffers
            monkeypatch.setenv("YOLORT_AMD_LB_BLOCKS", kernel.split("+b")[1])
    from yolort_amd.engine import View
    from yolort_amd.models.transform import YOLOTransform
    from workloads.synth import synth_images
    tr = YOLOTransform(320, 320)
    # first list: every image fits the LDS budget of the tiled kernels (scale factors <= 2.1: they run); second list: the 6x
    # down-scale of the 1080x1920 image does not (the whole launch falls back to the per-pixel kernel)
    for shapes in ([(641, 479), (97, 311), (320, 320), (333, 251), (75, 100), (500, 641), (640, 333)],
                   [(641, 479), (97, 311), (320, 320), (333, 251), (1080, 1920), (75, 100)]):
        imgs = []
        for i, (h, w) in enumerate(shapes):
            im = synth_images(1, h, w, seed=60 + i)[0]
            if in_dtype == torch.uint8:
                im = (im * 255).round().to(torch.uint8)
                im = im.permute(1, 2, 0).contiguous() if hwc else im
            else:
                im = im.to(in_dtype)
            imgs.append(im.to(dev))
        (hb, wb), sizes, pads = tr.geometry([tr.image_hw(im) for im in imgs])
        outs = []
        for c_out in (4, 8):
            t = torch.full((len(imgs) * hb * wb * c_out,), 7.0, device=dev, dtype=out_dtype)
            v = View(t, 0, len(imgs), hb, wb, c_out, c_out)
            tr.letterbox_into(imgs, v, sizes, pads)
            torch.cuda.synchronize()
            outs.append(v.as_tensor().clone())
        assert torch.equal(outs[0][..., :3], outs[1][..., :3])
        assert bool((outs[0][..., 3] == 0).all())


@pytest.mark.parametrize("hw,S", [((640, 640), 640), ((480, 640), 640), ((320, 256), 320)])
def test_letterbox_identity_fast_path(dev, hw, S):
    """batches whose images already have the resized size take the interleave-copy kernel: bit-identical to the
    bilinear kernel's result, which for scale 1 is the source pixel (oracle, fp32 exact)"""
    from oracle import yolov5_oracle as O
    from yolort_amd.models.transform import YOLOTransform
    from workloads.synth import synth_images
    imgs = [synth_images(1, hw[0], hw[1], seed=70 + i)[0] for i in range(3)]
    ref, sizes = O.letterbox(imgs, S, S, 32)
    assert sizes == [hw] * 3
    t = YOLOTransform(S, S)
    nt, _ = t([im.to(dev) for im in imgs], None, dtype=torch.float32)
    assert torch.equal(nt.nchw().cpu(), ref)
    nt16, _ = t([im.half().to(dev) for im in imgs], None, dtype=torch.float16)
    ref16, _ = O.letterbox([im.half().float() for im in imgs], S, S, 32)
    assert torch.equal(nt16.nchw().float().cpu(), ref16.half().float())
    u8 = [(im * 255).round().to(torch.uint8) for im in imgs]
    ref8, _ = O.letterbox([u.float() / 255.0 for u in u8], S, S, 32)
    nt8, _ = t([u.to(dev) for u in u8], None, dtype=torch.float32)
    assert torch.equal(nt8.nchw().cpu(), ref8)


def test_letterbox_interleaved_uint8_input(dev):
    """YMI_U8_HWC: decoded images (H, W, 3) uint8 go straight into the letterbox kernel -- the result must equal the planar
    uint8 path bit for bit (bilinear resize and identity sizes), i.e. permute + /255 + letterbox in one kernel"""
    from yolort_amd.models.transform import YOLOTransform
    from workloads.synth import synth_images
    for shapes, S in (([(1080, 810), (480, 640), (375, 500), (100, 37)], 640), ([(320, 256)] * 3, 320)):
        planar = [(synth_images(1, h, w, seed=h + w + i)[0] * 255).round().to(torch.uint8) for i, (h, w) in enumerate(shapes)]
        hwc = [u.permute(1, 2, 0).contiguous() for u in planar]
        t = YOLOTransform(S, S)
        assert all(t.is_hwc(u) for u in hwc) and not any(t.is_hwc(u) for u in planar)
        a, _ = t([u.to(dev) for u in planar], None, dtype=torch.float16)
        b, _ = t([u.to(dev) for u in hwc], None, dtype=torch.float16)
        assert a.image_sizes == b.image_sizes
        assert torch.equal(a.nchw(), b.nchw())
    with pytest.raises(Exception):
        t([planar[0].to(dev), hwc[1].to(dev)], None, dtype=torch.float16)   # mixed layouts in one batch


def _rand_boxes(rng, n, span=200.0):
    xy = rng.random((n, 2), dtype=np.float32) * span
    wh = rng.random((n, 2), dtype=np.float32) * 60 + 2
    return np.concatenate([xy, xy + wh], 1).astype(np.float32)


@pytest.mark.parametrize("n,ncls,ties", [(0, 3, False), (1, 1, False), (77, 3, True), (1000, 80, False), (5000, 5, True), (20000, 80, True)])
def test_batched_nms_bit_exact(dev, n, ncls, ties):
    from oracle import yolov5_oracle as O
    from yolort_amd.ops import batched_nms
    rng = np.random.default_rng(n + ncls)
    boxes = _rand_boxes(rng, n)
    scores = rng.random(n, dtype=np.float32)
    if ties:
        scores = np.round(scores, 2).astype(np.float32)
    labels = rng.integers(0, ncls, n).astype(np.int64)
    ref = O.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(labels), 0.45).numpy()
    got = batched_nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), torch.from_numpy(labels).to(dev), 0.45).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_batched_nms_takes_arbitrary_int64_category_ids(dev):
    """torchvision.ops.batched_nms accepts any int64 ids; the kernel's records hold 12 bits of class.  Ids >= 4096, negative ids and ids 2^40 apart must not alias:
    they are renumbered densely before the call, and more than 4096 distinct ids are refused"""
    from oracle import yolov5_oracle as O
    from yolort_amd._lib import YmiError
    from yolort_amd.ops import batched_nms
    rng = np.random.default_rng(7)
    n = 3000
    boxes, scores = _rand_boxes(rng, n, span=200), rng.random(n, dtype=np.float32)
    ids = np.array([-5, 0, 4095, 4096, 8191, 8192, 1 << 40, (1 << 40) + 4096], np.int64)   # 4096 / 8192 / 0 alias under a 12-bit mask, as do the last two
    labels = ids[rng.integers(0, len(ids), n)]
    dense = np.searchsorted(np.sort(ids), labels).astype(np.int64)
    ref = O.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(dense), 0.45).numpy()
    got = batched_nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), torch.from_numpy(labels).to(dev), 0.45).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    masked = O.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(labels & 4095), 0.45).numpy()
    assert len(masked) != len(ref)    # what aliasing would have returned differs on this input: the case is not vacuous
    with pytest.raises(YmiError):
        batched_nms(torch.from_numpy(_rand_boxes(rng, 5000)).to(dev), torch.rand(5000, device=dev), torch.arange(5000, device=dev), 0.45)


def test_postprocess_vs_oracle(dev):
    """decode + threshold + sort + NMS + top-k from the same fp32 head logits: integer outputs
    (labels, counts, order) bit-exact; boxes/scores to fp32 rounding of expf."""
    from oracle import yolov5_oracle as O
    from yolort_amd.ops import postprocess_logits
    g = torch.Generator().manual_seed(11)
    n, nc = 3, 80
    shapes = [(20, 24), (10, 12), (5, 6)]
    heads = [torch.randn(n, 3, h, w, nc + 5, generator=g) * 2.0 - 1.0 for h, w in shapes]
    strides, anchors = O.anchors_for(3)
    for thr, k in [(0.3, 300), (0.05, 50)]:
        pred = O.decode(heads, strides, anchors)
        ref = O.postprocess(pred, thr, 0.45, k)
        got = postprocess_logits([h.to(dev) for h in heads], strides, anchors, nc, thr, 0.45, k)
        for r, d in zip(ref, got):
            assert len(d["scores"]) == len(r["scores"])
            np.testing.assert_array_equal(d["labels"].cpu().numpy(), r["labels"].numpy())
            np.testing.assert_allclose(d["scores"].cpu().numpy(), r["scores"].numpy(), rtol=2e-6, atol=1e-7)
            np.testing.assert_allclose(d["boxes"].cpu().numpy(), r["boxes"].numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("aw,expect_short", [(30.0, False), (400.0, True)])
def test_postprocess_score_prefix_vs_oracle(dev, aw, expect_short):
    """A crowded image (> 6144 candidates) is cut to a score-ordered prefix before sort / NMS.  Small anchors: the
    prefix alone yields 300 survivors and the result must equal the oracle's full computation.  Huge anchors: nearly
    everything is suppressed, the prefix falls short (YMI_STATUS_PREFIX_SHORT) and the exact full pass must follow."""
    from oracle import yolov5_oracle as O
    from yolort_amd import _lib
    from yolort_amd.engine import Plan, View
    from yolort_amd.ops import postprocess_logits
    g = torch.Generator().manual_seed(31)
    nc = 4
    heads = [torch.randn(2, 3, 48, 48, nc + 5, generator=g)]
    heads[0][..., :4] *= 0.05                      # boxes ~ anchor sized, centred on their cells
    heads[0][1] -= 6.0                             # second image: almost nothing passes
    heads[0][0, ..., 4:] += 3.0                    # first image: nearly every (anchor, class) pair passes
    strides, anchors = [8], [[aw, aw, 1.1 * aw, 0.9 * aw, 0.9 * aw, 1.1 * aw]]
    ref = O.postprocess(O.decode(heads, strides, anchors), 0.3, 0.45, 300)
    got = postprocess_logits([h.to(dev) for h in heads], strides, anchors, nc, 0.3, 0.45, 300, cand_cap=2 * 32768)
    for r, d in zip(ref, got):
        np.testing.assert_array_equal(d["labels"].cpu().numpy(), r["labels"].numpy())
        np.testing.assert_allclose(d["scores"].cpu().numpy(), r["scores"].numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(d["boxes"].cpu().numpy(), r["boxes"].numpy(), rtol=1e-5, atol=1e-3)
    # the raw status of a single default pass shows which way it went
    t = torch.zeros(2, 48, 48, 28, device=dev)
    t[..., :27] = heads[0].to(dev).permute(0, 2, 3, 1, 4).reshape(2, 48, 48, 27)
    plan = Plan(dev, torch.float16)
    pb = plan.postprocess([View(t.view(-1), 0, 2, 48, 48, 27, 28)], strides, anchors, nc, 0.3, 0.45, 300, 2 * 32768)
    plan.run()
    st = pb.status.cpu().tolist()
    assert st[0] < 3 * 48 * 48 * nc, "the first image should have been cut to a prefix"
    assert bool(st[1] & 2) == expect_short and not st[1] & 1


def test_postprocess_overflow_is_reported_and_recovered(dev):
    from oracle import yolov5_oracle as O
    from yolort_amd.ops import postprocess_logits
    g = torch.Generator().manual_seed(12)
    heads = [torch.randn(1, 3, 8, 8, 85, generator=g) + 3.0]
    strides, anchors = [8], [O.ANCHORS_P5[0]]
    ref = O.postprocess(O.decode(heads, strides, anchors), 0.3, 0.45, 300)
    got = postprocess_logits([h.to(dev) for h in heads], strides, anchors, 80, 0.3, 0.45, 300, cand_cap=64)  # far too small -> grows
    np.testing.assert_array_equal(got[0]["labels"].cpu().numpy(), ref[0]["labels"].numpy())
