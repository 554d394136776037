"""Shared by tests/test_hipsim_model.py (CPU simulator) and tests/test_reference_blocks_gpu.py: the block-level cases of the reference's own test/test_v5_common.py
(TestConv :41-75, TestBottleneck :85-96, TestC3 :99-110, TestSPP :113-118, TestSPPF :121-126, TestFocus :129-134) -- same constructor calls, same input shapes -- plus
the r3.1 block the reference has no unit test for (BottleneckCSP), each with the oracle's fp32 evaluation of the same module as the expected VALUE (the reference's
tests assert shapes only)."""
import torch
import torch.nn.functional as F


def cases():
    from yolort_amd.v5 import C3, SPP, SPPF, Bottleneck, BottleneckCSP, Conv, Focus
    return {
        "Conv.output_shape": (lambda: Conv(3, 16, k=3, s=1), (1, 3, 32, 32), (1, 16, 32, 32)),
        "Conv.stride_2": (lambda: Conv(3, 16, k=3, s=2), (1, 3, 32, 32), (1, 16, 16, 16)),
        "Conv.version_r31": (lambda: Conv(3, 16, k=3, s=1, version="r3.1"), (1, 3, 32, 32), (1, 16, 32, 32)),
        "Conv.no_activation": (lambda: Conv(3, 16, k=1, act=False), (1, 3, 32, 32), (1, 16, 32, 32)),
        "Bottleneck.with_shortcut": (lambda: Bottleneck(64, 64, shortcut=True), (1, 64, 16, 16), (1, 64, 16, 16)),
        "Bottleneck.without_shortcut": (lambda: Bottleneck(64, 32, shortcut=False), (1, 64, 16, 16), (1, 32, 16, 16)),
        "C3.output_shape": (lambda: C3(64, 64, n=1), (1, 64, 16, 16), (1, 64, 16, 16)),
        "C3.different_channels": (lambda: C3(32, 64, n=2), (1, 32, 16, 16), (1, 64, 16, 16)),
        "SPP.output_shape": (lambda: SPP(64, 64), (1, 64, 16, 16), (1, 64, 16, 16)),
        "SPPF.output_shape": (lambda: SPPF(64, 64), (1, 64, 16, 16), (1, 64, 16, 16)),
        "Focus.output_shape": (lambda: Focus(3, 64, k=3), (1, 3, 64, 64), (1, 64, 32, 32)),
        "Focus.r31": (lambda: Focus(3, 32, k=3, version="r3.1"), (2, 3, 32, 48), (2, 32, 16, 24)),
        "BottleneckCSP.r31": (lambda: BottleneckCSP(64, 64, n=2), (1, 64, 16, 16), (1, 64, 16, 16)),
    }


def build(name, seed=0):
    """the module in eval mode with non-trivial BatchNorm statistics, and a seeded input of the reference test's shape"""
    make, in_shape, out_shape = cases()[name]
    torch.manual_seed(seed)
    m = make().eval()
    g = torch.Generator().manual_seed(100 + seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 0.8 + 0.6)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.2)
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.3)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    x = torch.randn(in_shape, generator=g)
    return m, x, out_shape


def expected(name, m, x):
    """fp32 value of the block through the oracle's restatement of the reference's module (oracle/yolov5_oracle.py)"""
    from oracle import yolov5_oracle as O
    sd = {"m." + k: v.float() for k, v in m.state_dict().items()}
    kind = type(m).__name__
    r31 = name.endswith("r31")
    O.VERSION.act = "hardswish" if r31 else "silu"
    try:
        with torch.no_grad():
            if kind == "Conv":
                if name == "Conv.no_activation":
                    y = F.conv2d(x, sd["m.conv.weight"], None, m.conv.stride, m.conv.padding)
                    return F.batch_norm(y, sd["m.bn.running_mean"], sd["m.bn.running_var"], sd["m.bn.weight"], sd["m.bn.bias"], False, 0.0, m.bn.eps)
                return O.conv_bn_silu(x, sd, "m", stride=m.conv.stride[0])
            if kind == "Bottleneck":
                z = O.conv_bn_silu(O.conv_bn_silu(x, sd, "m.cv1"), sd, "m.cv2")
                return x + z if m.add else z
            if kind == "C3":
                return O.c3(x, sd, "m", shortcut=True)
            if kind in ("SPP", "SPPF"):
                return O.spp(x, sd, "m")
            if kind == "Focus":
                return O.focus(x, sd, "m")
            if kind == "BottleneckCSP":
                return O.bottleneck_csp(x, sd, "m", shortcut=True)
    finally:
        O.VERSION.act = "silu"
    raise KeyError(name)
