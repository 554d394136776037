"""Detection head and post-process (reference yolort/models/box_head.py).

YOLOHead  : one biased 1x1 conv per level (box_head.py:14-82).  Emitted as the same fused
            implicit-GEMM kernel with fp32 output, NHWC (N,H,W,A*K): the reference's
            view/permute/contiguous round trip (:76-78) never happens on the fused path.
PostProcess: sigmoid + anchor decode + multi-label threshold + class-aware NMS + top-k
            (box_head.py:328-360, :363-429) as ONE recorded op (csrc/postprocess.hip).
The training criterion `SetCriterion` (box_head.py:85-325) is out of scope (inference path only).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor, nn

from .._lib import ACT_NONE, YmiError
from ..engine import PackedConv, Plan, View
from ..hipmodule import HipModule, compute_dtype_of


class YOLOHead(HipModule):
    def __init__(self, in_channels: List[int], num_anchors: int, strides: List[int], num_classes: int):
        super().__init__()
        if not isinstance(in_channels, list):
            in_channels = [in_channels] * len(strides)
        self.num_anchors = num_anchors
        self.num_classes = num_classes
        self.num_outputs = num_classes + 5
        self.strides = strides
        blocks = nn.ModuleList(nn.Conv2d(ch, self.num_outputs * self.num_anchors, 1) for ch in in_channels)
        # bias prior of the reference (box_head.py:40-46): 8 objects per 640 image, class prior 0.6/nc
        for mi, s in zip(blocks, self.strides):
            b = mi.bias.detach().view(self.num_anchors, -1).clone()
            b[:, 4] += math.log(8 / (640 / s) ** 2)
            b[:, 5:] += math.log(0.6 / (self.num_classes - 0.999999))
            mi.bias = nn.Parameter(b.view(-1), requires_grad=True)
        self.head = blocks
        self._packed: Dict = {}
        self.group_levels = os.environ.get("YOLORT_AMD_HEAD_GROUP", "1") != "0"   # fused heads of all levels in one launch

    def packed(self, i: int, dtype, device, cin_view: int) -> PackedConv:
        m = self.head[i]
        sig = (m.weight._version, m.bias._version, m.weight.data_ptr())
        key = (i, dtype, device, cin_view)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        pc = PackedConv(m.weight, m.bias, None, dtype, device, cin_pad=cin_view)
        self._packed[key] = (sig, pc)
        return pc

    def can_fuse_decode(self, plan: Plan, x: Sequence[View]) -> bool:
        """the fused head needs <= 128 outputs per anchor, 3 anchors, 32-aligned input channels and the pipelined kernels"""
        return (self.num_anchors == 3 and self.num_outputs <= 128 and not plan.use_v1 and not plan.fp32 and all(f.c % 32 == 0 and f.tail >= 0 and f.h * f.w >= 1 for f in x))

    def packed_anchor_major(self, i: int, dtype, device, cin_view: int) -> PackedConv:
        """head weights for ymi_conv_head_decode: anchor q's K rows at q*RA .. q*RA+K-1, RA = round_up(K, 32), rest zero"""
        m = self.head[i]
        sig = (m.weight._version, m.bias._version, m.weight.data_ptr())
        key = ("fused", i, dtype, device, cin_view)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        k = self.num_outputs
        ra = (k + 31) // 32 * 32
        w = m.weight.detach()
        wp = torch.zeros(3 * ra, w.shape[1], 1, 1, dtype=w.dtype, device=w.device)
        bp = torch.zeros(3 * ra, dtype=m.bias.dtype, device=w.device)
        for q in range(3):
            wp[q * ra: q * ra + k] = w[q * k: (q + 1) * k]
            bp[q * ra: q * ra + k] = m.bias.detach()[q * k: (q + 1) * k]
        pc = PackedConv(wp, bp, None, dtype, device, cin_pad=cin_view)
        pc.k_real = w.shape[1]
        self._packed[key] = (sig, pc)
        return pc

    def emit_fused(self, plan: Plan, x: Sequence[View], post_desc) -> None:
        """head conv + decode + threshold in one kernel per level (csrc/head_decode.hpp); nothing is returned: boxes and
        candidate records land in the post-process workspace of `post_desc`"""
        pcs = [self.packed_anchor_major(i, plan.dtype, plan.device, f.c) for i, f in enumerate(x)]
        if self.group_levels and len(x) > 1:
            plan.head_decode_group(list(x), pcs, post_desc, name="head.*")   # all levels in one launch
            return
        for i, f in enumerate(x):
            plan.head_decode(f, pcs[i], post_desc, i, name=f"head.{i}")

    def emit(self, plan: Plan, x: Sequence[View], out=None) -> List[View]:
        """returns fp32 logits views (N,H,W,A*K) with channel a*K + k"""
        outs = []
        for i, f in enumerate(x):
            outs.append(plan.conv(f, self.packed(i, plan.dtype, plan.device, f.c), 1, 0, ACT_NONE, out_dtype=torch.float32, name=f"head.{i}"))
        return outs

    def forward(self, x: List[Tensor]) -> List[Tensor]:
        """API parity: List[(N,C,H,W)] -> List[(N,A,H,W,K)] (box_head.py:68-82)."""
        ys = super().forward(list(x))  # NCHW (N, A*K, H, W)
        outs = []
        for y in ys:
            n, _, h, w = y.shape
            outs.append(y.view(n, self.num_anchors, -1, h, w).permute(0, 1, 3, 4, 2).contiguous())
        return outs


class PostProcess(nn.Module):
    """Same constructor as the reference (box_head.py:376-389)."""

    def __init__(self, strides: List[int], score_thresh: float, nms_thresh: float, detections_per_img: int) -> None:
        super().__init__()
        self.strides = strides
        self.score_thresh = score_thresh
        self.nms_thresh = nms_thresh
        self.detections_per_img = detections_per_img

    def forward(self, head_outputs: List[Tensor], grids: List[Tensor], shifts: List[Tensor]) -> List[Dict[str, Tensor]]:
        """Stand-alone use with the reference calling convention: head_outputs[l] is (N,A,H,W,K);
        anchors are read back from `shifts` (they are constant over H,W, anchor_utils.py:49-59)."""
        from ..ops import postprocess_logits

        anchors = []
        for s in shifts:
            a = s[0, :, 0, 0, :].reshape(-1).float().cpu().tolist()
            anchors.append(a)
        k = head_outputs[0].shape[-1]
        return postprocess_logits(head_outputs, [float(s) for s in self.strides], anchors, k - 5, self.score_thresh, self.nms_thresh, self.detections_per_img)
