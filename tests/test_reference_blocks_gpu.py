"""The reference's own block-level tests against this package's modules on the MI355X ("the reference's per-block shape tests run unchanged"):
test/test_v5_common.py TestConv / TestBottleneck / TestC3 / TestSPP / TestSPPF / TestFocus and test/test_models.py::test_backbone_with_pan (:188-224) -- same constructor
calls and input shapes (tests/_blocks.py), called the way the reference calls them (`module(x)` on an NCHW tensor: HipModule.forward converts at the edge and runs the
module's own plan).  The reference asserts output shapes; here the VALUE is checked too, against the oracle's fp32 evaluation of the same module."""
import pytest
import torch

import _blocks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from yolort_amd import _lib
    _lib.load(require_gpu=True)
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", sorted(_blocks.cases()))
def test_v5_common_blocks(dev, name):
    m, x, out_shape = _blocks.build(name)
    want = _blocks.expected(name, m, x)
    out = m.to(dev).half()(x.to(dev).half())
    assert tuple(out.shape) == out_shape and out.dtype == torch.float16 and out.is_cuda
    got = out.float().cpu()
    assert (got - want).abs().max().item() <= 1e-2 * max(1.0, want.abs().max().item()), name
    out2 = m(x.to(dev).half())          # the cached plan replays
    assert torch.equal(out, out2)


def test_v5_common_conv_invalid_version():
    from yolort_amd.v5 import Conv
    with pytest.raises(NotImplementedError):
        Conv(3, 16, k=3, s=1, version="r99.0")   # reference test_v5_common.py:60-62


@pytest.mark.parametrize("depth_multiple,width_multiple,version,use_p6", [(0.33, 0.5, "r3.1", False), (0.33, 0.5, "r4.0", False), (0.33, 0.5, "r6.0", False), (0.33, 0.5, "r6.0", True)])
def test_backbone_with_pan(dev, depth_multiple, width_multiple, version, use_p6):
    """reference test/test_models.py:188-224 (`_get_feature_shapes` :66-79): strides 8 / 16 / 32 (/ 64), channels 256 / 512 / (768) / 1024 x width_multiple"""
    from yolort_amd.models.backbone_utils import darknet_pan_backbone
    n, h, w = 4, 448, 320
    model = darknet_pan_backbone(f"darknet_s_{version.replace('.', '_')}", depth_multiple, width_multiple, version=version, use_p6=use_p6).to(dev).half().eval()
    out = model(torch.rand(n, 3, h, w, device=dev).half())
    strides = [8, 16, 32, 64] if use_p6 else [8, 16, 32]
    widths = [256, 512, 768, 1024] if use_p6 else [256, 512, 1024]
    assert len(out) == len(strides)
    for o, s, c in zip(out, strides, widths):
        assert tuple(o.shape) == (n, int(c * width_multiple), h // s, w // s)
        assert bool(torch.isfinite(o.float()).all())
