"""Backbone glue (reference yolort/models/backbone_utils.py:11-122): DarkNet `.features` truncated
at layer 8 with taps after layers 4/6/8, plus the PAN.  The taps are written by their producing
C3 directly into the PAN's top-down concat buffers."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

from torch import nn

from ..engine import Plan, View
from ..hipmodule import HipModule
from . import darknetv4, darknetv6
from .path_aggregation_network import PathAggregationNetwork


class Body(nn.ModuleDict):
    """Children "0".."k" of the backbone up to the last requested layer; same key naming as
    torchvision's IntermediateLayerGetter so checkpoints keep `body.{i}.` prefixes."""

    def __init__(self, model: nn.Module, return_layers: Dict[str, str]) -> None:
        names = [name for name, _ in model.named_children()]
        if not set(return_layers).issubset(names):
            raise ValueError("return_layers are not present in model")
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)


class BackboneWithPAN(HipModule):
    def __init__(self, backbone, return_layers, in_channels_list, depth_multiple, version, use_p6=False):
        super().__init__()
        self.body = Body(backbone, return_layers={str(k): str(v) for k, v in return_layers.items()})
        self.pan = PathAggregationNetwork(in_channels_list, depth_multiple, version=version, use_p6=use_p6)
        self.out_channels = in_channels_list

    def _input_cpad(self, c: int) -> int:
        return 4 if c == 3 else (c + 7) // 8 * 8

    def emit(self, plan: Plan, x: View, out=None, name: str = "backbone") -> List[View]:
        nf = len(self.out_channels)
        taps: List[View] = []
        td_cat: Dict[int, View] = {}
        for lname, layer in self.body.items():
            ti = self.body.return_layers.get(lname)
            if ti is None:
                x = layer.emit(plan, x, name=f"{name}.body.{lname}")
                continue
            step = nf - 2 - int(ti)  # top-down step that concatenates this tap (reference pan :224)
            if step >= 0:
                c_up = self.pan.td_slot_channels(step)
                c_tap = layer.out_channels
                cat = plan.alloc(x.n, x.h, x.w, c_up + c_tap)
                td_cat[step] = cat
                x = layer.emit(plan, x, out=cat.slice_c(c_up, c_tap), name=f"{name}.body.{lname}")
            else:
                x = layer.emit(plan, x, name=f"{name}.body.{lname}")
            taps.append(x)
        return self.pan.emit(plan, taps, td_cat=td_cat, name=f"{name}.pan")


def darknet_pan_backbone(backbone_name: str, depth_multiple: float, width_multiple: float, pretrained: Optional[bool] = False,
                         returned_layers: Optional[List[int]] = None, version: str = "r6.0", use_p6: bool = False):
    """Same signature as the reference (:60-122)."""
    assert version in ["r3.1", "r4.0", "r6.0"], "Currently only supports version 'r3.1', 'r4.0' and 'r6.0'."
    last_channel = 768 if use_p6 else 1024
    factories = {**darknetv4.__dict__, **darknetv6.__dict__}   # the reference's `darknet` namespace (darknet.py re-exports both)
    backbone = factories[backbone_name](pretrained=pretrained, last_channel=last_channel).features
    if returned_layers is None:
        returned_layers = [4, 6, 8]
    return_layers = {str(k): str(i) for i, k in enumerate(returned_layers)}
    grow_widths = [256, 512, 768, 1024] if use_p6 else [256, 512, 1024]
    in_channels_list = [int(gw * width_multiple) for gw in grow_widths]
    return BackboneWithPAN(backbone, return_layers, in_channels_list, depth_multiple, version, use_p6=use_p6)
