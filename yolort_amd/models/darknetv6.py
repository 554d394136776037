"""CSPDarknet r6.0 body (reference yolort/models/darknetv6.py:31-199).

Only `.features` (stem Conv(3,c,k=6,s=2,p=2) + 4 x [Conv k3 s2, C3]) is on the inference path; the
reference's ImageNet classifier head (avgpool + classifier, :98-105) is kept as plain parameter
containers so `state_dict()` of a full DarkNetV6 matches, but it is never emitted.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional

from torch import nn

from ..engine import Plan, View
from ..hipmodule import HipModule
from ..v5 import C3, Conv
from ._utils import _make_divisible

__all__ = ["DarkNetV6", "darknet_n_r6_0", "darknet_s_r6_0", "darknet_m_r6_0", "darknet_l_r6_0", "darknet_x_r6_0"]


class Features(HipModule, nn.Sequential):
    """nn.Sequential of the body layers that can emit itself (taps optional)."""

    def __init__(self, *layers: nn.Module) -> None:
        nn.Sequential.__init__(self, *layers)
        self._plans = {}

    def _input_cpad(self, c: int) -> int:
        return 4 if c == 3 else (c + 7) // 8 * 8

    def emit(self, plan: Plan, x: View, out=None, name: str = "features") -> View:
        for i, layer in enumerate(self):
            x = layer.emit(plan, x, name=f"{name}.{i}")
        return x

    def forward(self, x):
        return HipModule.forward(self, x)


class DarkNetV6(nn.Module):
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
        assert version == "r4.0", "Currently the module version used in DarkNetV6 is r4.0."
        block = block or C3
        stages_repeats = stages_repeats or [3, 6, 9]
        stages_out_channels = stages_out_channels or [128, 256, 512]

        width = _make_divisible(64 * width_multiple, round_nearest)
        layers: List[nn.Module] = [Conv(3, width, k=6, s=2, p=2, version=version)]  # reference :81
        for repeats, channels in zip(stages_repeats, stages_out_channels):  # reference :85-90
            n = max(round(repeats * depth_multiple), 1)
            c = _make_divisible(channels * width_multiple, round_nearest)
            layers += [Conv(width, c, k=3, s=2, version=version), block(c, c, n=n)]
            width = c
        last = _make_divisible(last_channel * width_multiple, round_nearest)  # reference :93-96
        layers += [Conv(width, last, k=3, s=2, version=version), block(last, last, n=max(round(3 * depth_multiple), 1))]
        self.features = Features(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Linear(last, last), nn.Hardswish(inplace=True), nn.Dropout(p=0.2, inplace=True), nn.Linear(last, num_classes))
        for m in self.modules():  # reference :107-114
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    def forward(self, x):
        raise NotImplementedError("the ImageNet classifier of DarkNetV6 is not part of the YOLOv5 inference path; use `.features`")


def _darknet_v6_conf(arch: str, pretrained: bool, progress: bool, *args: Any, **kwargs: Any) -> DarkNetV6:
    if pretrained:
        raise NotImplementedError(f"pretrained {arch} is not supported as of now")  # same as reference :137-139
    return DarkNetV6(*args, **kwargs)


def darknet_n_r6_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV6:
    return _darknet_v6_conf("darknet_n_r6.0", pretrained, progress, 0.33, 0.25, **kwargs)


def darknet_s_r6_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV6:
    return _darknet_v6_conf("darknet_s_r6.0", pretrained, progress, 0.33, 0.5, **kwargs)


def darknet_m_r6_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV6:
    return _darknet_v6_conf("darknet_m_r6.0", pretrained, progress, 0.67, 0.75, **kwargs)


def darknet_l_r6_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV6:
    return _darknet_v6_conf("darknet_l_r6.0", pretrained, progress, 1.0, 1.0, **kwargs)


def darknet_x_r6_0(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> DarkNetV6:
    return _darknet_v6_conf("darknet_x_r6.0", pretrained, progress, 1.33, 1.25, **kwargs)
