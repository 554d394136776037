"""PAN neck (reference yolort/models/path_aggregation_network.py:10-245): r6.0 (SPP first, C3 blocks), r4.0 (C3 first, C3 blocks) and r3.1 (BottleneckCSP throughout,
Hardswish convolutions).

Top-down: SPP / block -> 1x1 Conv -> nearest x2 upsample -> concat with the backbone tap -> C3;
bottom-up: 3x3 s2 Conv -> concat with the matching top-down tensor -> C3; optional P6 level.
Every concat of the reference (:224, :235) is a pre-allocated buffer whose halves are written in
place by their producers (1x1 conv epilogue, upsample kernel, backbone tap, 3x3 s2 conv).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

from torch import nn

from ..engine import Plan, View
from ..hipmodule import HipModule
from ..v5 import C3, SPP, BottleneckCSP, Conv

_block = {"r3.1": BottleneckCSP, "r4.0": C3}   # reference :242-245


class IntermediateLevelP6(HipModule):
    """Extra stride-64 stage appended to the feature list (reference :10-41)."""

    def __init__(self, depth_multiple: float, in_channel: int, out_channel: int, version: str = "r4.0"):
        super().__init__()
        n = max(round(3 * depth_multiple), 1)
        self.p6 = nn.Sequential(Conv(in_channel, out_channel, k=3, s=2, version=version), _block[version](out_channel, out_channel, n=n))

    def emit(self, plan: Plan, x, out=None, name: str = "p6"):
        feats = list(x) if isinstance(x, (list, tuple)) else [x]
        y = self.p6[0].emit(plan, feats[-1], name=name + ".0")
        feats.append(self.p6[1].emit(plan, y, name=name + ".1"))
        return feats


class PathAggregationNetwork(HipModule):
    def __init__(self, in_channels: List[int], depth_multiple: float, version: str = "r4.0",
                 block: Optional[Callable[..., nn.Module]] = None, use_p6: bool = False):
        super().__init__()
        if version not in ("r3.1", "r4.0", "r6.0"):
            raise NotImplementedError(f"Version {version} is not implemented yet.")
        mv = "r4.0" if version == "r6.0" else version  # module_version of the reference (:87)
        if use_p6:
            assert len(in_channels) == 4, "Length of in channels should be 4."
            self.intermediate_blocks = IntermediateLevelP6(depth_multiple, in_channels[2], in_channels[3], version=mv)
        else:
            assert len(in_channels) == 3, "Length of in channels should be 3."
            self.intermediate_blocks = None
        block = block or _block[mv]
        n = max(round(3 * depth_multiple), 1)
        c = in_channels
        # reference :109-114: the r6.0 neck opens with the SPP (its backbone ends in a C3), the older ones with a block (their backbones end in the SPP)
        inner: List[nn.Module] = [SPP(c[-1], c[-1], k=(5, 9, 13))] if version == "r6.0" else [block(c[-1], c[-1], n=n, shortcut=False)]
        if use_p6:
            inner += [Conv(c[-1], c[2], 1, 1, version=mv), nn.Upsample(scale_factor=2), block(c[1] + c[-1], c[2], n=n, shortcut=False)]
        inner += [Conv(c[2], c[1], 1, 1, version=mv), nn.Upsample(scale_factor=2), block(c[-1], c[1], n=n, shortcut=False),
                  Conv(c[1], c[0], 1, 1, version=mv), nn.Upsample(scale_factor=2)]
        self.inner_blocks = nn.ModuleList(inner)
        layer: List[nn.Module] = [block(c[1], c[0], n=n, shortcut=False), Conv(c[0], c[0], 3, 2, version=mv),
                                  block(c[1], c[1], n=n, shortcut=False), Conv(c[1], c[1], 3, 2, version=mv),
                                  block(c[-1], c[2], n=n, shortcut=False)]
        if use_p6:
            layer += [Conv(c[2], c[2], 3, 2, version=mv), block(c[1] + c[-1], c[-1], n=n, shortcut=False)]
        self.layer_blocks = nn.ModuleList(layer)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    def td_slot_channels(self, step: int) -> int:
        """channels of the upsampled tensor concatenated in top-down step `step`"""
        return self.inner_blocks[3 * step + 1].conv.out_channels

    def emit(self, plan: Plan, x, out=None, td_cat: Optional[Dict[int, View]] = None, name: str = "pan") -> List[View]:
        feats: List[View] = list(x.values()) if isinstance(x, dict) else list(x)
        if self.intermediate_blocks is not None:
            feats = self.intermediate_blocks.emit(plan, feats, name=name + ".intermediate_blocks.p6")
        nf = len(feats)
        n = feats[0].n
        bu_cat: Dict[int, View] = {}
        last = feats[-1]
        for t in range(nf - 1):  # reference :219-224
            last = self.inner_blocks[3 * t].emit(plan, last, name=f"{name}.inner_blocks.{3 * t}")
            level, idx = nf - 1 - t, nf - 2 - t
            conv = self.inner_blocks[3 * t + 1]
            c_t = conv.conv.out_channels
            c_down = self.layer_blocks[2 * idx + 1].conv.out_channels
            bu_cat[idx] = plan.alloc(n, feats[level].h, feats[level].w, c_down + c_t)
            tap = feats[level - 1]
            if td_cat is not None and t in td_cat:
                cat = td_cat[t]
            else:
                cat = plan.alloc(n, tap.h, tap.w, c_t + tap.c)
                plan.copy(tap, cat.slice_c(c_t, tap.c), name=f"{name}.cat_tap.{t}")
            # nn.Upsample(scale_factor=2) (reference :221-223): folded into the 1x1 conv's epilogue when its channel count
            # allows (the conv writes its output AND the four upsampled copies), else a separate kernel
            fold = (c_t % 32 == 0 and not plan.use_v1 and (cat.h, cat.w) == (2 * feats[level].h, 2 * feats[level].w)
                    and isinstance(conv.act, (nn.SiLU, nn.Identity)))   # (an r3.1 Hardswish layer is two launches -- conv + ymi_act: its upsample stays a launch of its own)
            top = conv.emit(plan, last, out=bu_cat[idx].slice_c(c_down, c_t), name=f"{name}.inner_blocks.{3 * t + 1}",
                            up2_out=cat.slice_c(0, c_t) if fold else None)
            if not fold:
                plan.upsample2x(top, cat.slice_c(0, c_t), name=f"{name}.inner_blocks.{3 * t + 2}")
            last = cat
        results = [self.layer_blocks[0].emit(plan, last, name=f"{name}.layer_blocks.0")]  # reference :230-231
        for idx in range(nf - 1):  # reference :233-237
            conv = self.layer_blocks[2 * idx + 1]
            conv.emit(plan, results[-1], out=bu_cat[idx].slice_c(0, conv.conv.out_channels), name=f"{name}.layer_blocks.{2 * idx + 1}")
            results.append(self.layer_blocks[2 * idx + 2].emit(plan, bu_cat[idx], name=f"{name}.layer_blocks.{2 * idx + 2}"))
        return results
