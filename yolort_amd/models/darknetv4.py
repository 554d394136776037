"""CSPDarknet of the r3.1 / r4.0 releases (reference yolort/models/darknetv4.py:32-219): Focus stem + 3 x [Conv k3 s2, block(n)] + Conv k3 s2 + SPP, block =
BottleneckCSP (r3.1: Hardswish / LeakyReLU) or C3 (r4.0: SiLU).

Only `.features` is on the inference path; the ImageNet classifier head (:99-105) is kept as plain parameter containers so `state_dict()` of a full DarkNetV4 matches the
reference's, but it is never emitted.  The Focus stem runs as the 6 x 6 stride-2 convolution it is equivalent to (v5/models/common.py Focus): every stem kernel of the
r6.0 path applies unchanged.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional

from torch import nn

from ..v5 import C3, SPP, BottleneckCSP, Conv, Focus
from ._utils import _make_divisible
from .darknetv6 import Features

__all__ = ["DarkNetV4", "darknet_s_r3_1", "darknet_m_r3_1", "darknet_l_r3_1", "darknet_s_r4_0", "darknet_m_r4_0", "darknet_l_r4_0"]

_block = {"r3.1": BottleneckCSP, "r4.0": C3}   # reference :133-136


class DarkNetV4(nn.Module):
    def __init__(
        self,
        depth_multiple: float,
        width_multiple: float,
        version: str = "r4.0",
        block: Optional[Callable[..., nn.Module]] = None,
        stages_repeats: Optional[List[int]] = None,
        stages_out_channels: Optional[List[int]] = None,
        num_classes: int = 1000,
        round_nearest: int = 8,
        last_channel: int = 1024,
    ) -> None:
        super().__init__()
        assert version in ["r3.1", "r4.0"], "Currently the module version used in DarkNetV4 is r3.1 or r4.0"
        block = block or _block[version]
        stages_repeats = stages_repeats or [3, 9, 9]
        stages_out_channels = stages_out_channels or [128, 256, 512]

        width = _make_divisible(64 * width_multiple, round_nearest)
        layers: List[nn.Module] = [Focus(3, width, k=3, version=version)]  # reference :86
        for repeats, channels in zip(stages_repeats, stages_out_channels):  # reference :90-95
            n = max(round(repeats * depth_multiple), 1)
            c = _make_divisible(channels * width_multiple, round_nearest)
            layers += [Conv(width, c, k=3, s=2, version=version), block(c, c, n=n)]
            width = c
        last = _make_divisible(last_channel * width_multiple, round_nearest)  # reference :98-100
        layers += [Conv(width, last, k=3, s=2, version=version), SPP(last, last, k=(5, 9, 13), version=version)]
        self.features = Features(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Linear(last, last), nn.Hardswish(inplace=True), nn.Dropout(p=0.2, inplace=True), nn.Linear(last, num_classes))
        for m in self.modules():  # reference :107-115
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    def forward(self, x):
        raise NotImplementedError("the ImageNet classifier of DarkNetV4 is not part of the YOLOv5 inference path; use `.features`")


def _darknet_v4_conf(arch: str, pretrained: bool, progress: bool, *args: Any, **kwargs: Any) -> DarkNetV4:
    if pretrained:
        raise NotImplementedError(f"pretrained {arch} is not supported as of now")  # same as reference :146-149
    return DarkNetV4(*args, **kwargs)


def darknet_s_r3_1(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV4:
    return _darknet_v4_conf("darknet_s_r3.1", pretrained, progress, 0.33, 0.5, version="r3.1", **kwargs)


def darknet_m_r3_1(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV4:
    return _darknet_v4_conf("darknet_m_r3.1", pretrained, progress, 0.67, 0.75, version="r3.1", **kwargs)


def darknet_l_r3_1(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV4:
    return _darknet_v4_conf("darknet_l_r3.1", pretrained, progress, 1.0, 1.0, version="r3.1", **kwargs)


def darknet_s_r4_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV4:
    return _darknet_v4_conf("darknet_s_r4.0", pretrained, progress, 0.33, 0.5, version="r4.0", **kwargs)


def darknet_m_r4_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV4:
    return _darknet_v4_conf("darknet_m_r4.0", pretrained, progress, 0.67, 0.75, version="r4.0", **kwargs)


def darknet_l_r4_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV4:
    return _darknet_v4_conf("darknet_l_r4.0", pretrained, progress, 1.0, 1.0, version="r4.0", **kwargs)
