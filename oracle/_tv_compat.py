"""TEST INFRASTRUCTURE ONLY -- stand-in for the `torchvision` symbols the reference touches.

The reference (zhiqwang/yolort) imports torchvision, which is not installed in this image
(SURVEY.md section 0).  This module restates, from torchvision's published semantics, exactly the
seven symbols on the reference's hot path so that the UNMODIFIED reference
under /root/reference can be imported read-only and executed on CPU to pin the oracle:

  torchvision._is_tracing                          used at yolort/models/transform.py:59,70,302
  torchvision.ops.box_convert                      used at yolort/models/box_head.py:358
  torchvision.ops.boxes.batched_nms / nms          used at yolort/models/box_head.py:422
  torchvision.models._utils.IntermediateLayerGetter used at yolort/models/backbone_utils.py:45
  torchvision.io.read_image / ImageReadMode        used at yolort/models/yolov5.py:9,228

PARITY UNPINNED for batched_nms: torchvision's C++ kernel is absent, so the NMS contract is the
one fixed in SURVEY.md Appendix C-4 (stable score-descending order, strict `>` IoU test, per-class
exact form for every size, fp32 IoU).  Nothing in the product package may import this file.
"""
from __future__ import annotations

import sys
import types
from collections import OrderedDict
from typing import Dict

import torch
from torch import Tensor, nn


def _is_tracing() -> bool:
    return bool(torch._C._get_tracing_state())


def box_convert(boxes: Tensor, in_fmt: str, out_fmt: str) -> Tensor:
    if in_fmt == out_fmt:
        return boxes.clone()
    if in_fmt == "cxcywh" and out_fmt == "xyxy":
        cx, cy, w, h = boxes.unbind(-1)
        return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)
    if in_fmt == "xyxy" and out_fmt == "cxcywh":
        x1, y1, x2, y2 = boxes.unbind(-1)
        return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), dim=-1)
    raise ValueError(f"unsupported conversion {in_fmt}->{out_fmt}")


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Greedy NMS: stable score-descending order, suppress when IoU > threshold (strict)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    b = boxes.detach().to(torch.float32).cpu().numpy()
    s = scores.detach().to(torch.float32).cpu()
    order = torch.sort(s, descending=True, stable=True).indices.numpy()
    import numpy as np

    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    for _i in range(n):
        if suppressed[_i]:
            continue
        i = order[_i]
        keep.append(i)
        rest = order[_i + 1 :]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[_i + 1 :] |= ovr > thr
    return torch.as_tensor(np.asarray(keep, dtype=np.int64), device=boxes.device)


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float) -> Tensor:
    """Class-aware NMS, per-class exact form for all sizes."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        curr = torch.where(idxs == class_id)[0]
        k = nms(boxes[curr], scores[curr], iou_threshold)
        keep_mask[curr[k]] = True
    keep = torch.where(keep_mask)[0]
    return keep[torch.sort(scores[keep], descending=True, stable=True).indices]


class IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model: nn.Module, return_layers: Dict[str, str]) -> None:
        if not set(return_layers).issubset([name for name, _ in model.named_children()]):
            raise ValueError("return_layers are not present in model")
        orig = return_layers
        return_layers = {str(k): str(v) for k, v in return_layers.items()}
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            if name in return_layers:
                del return_layers[name]
            if not return_layers:
                break
        super().__init__(layers)
        self.return_layers = {str(k): str(v) for k, v in orig.items()}

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class ImageReadMode:
    UNCHANGED = 0
    GRAY = 1
    RGB = 3


def read_image(path: str, mode=ImageReadMode.RGB) -> Tensor:
    import numpy as np
    from PIL import Image

    img = Image.open(path)
    if mode == ImageReadMode.RGB:
        img = img.convert("RGB")
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()


def install() -> None:
    """Register the stand-in as sys.modules['torchvision'] (oracle harness only)."""
    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_ymi_stand_in", False):
        return  # a real torchvision exists: use it
    tv = types.ModuleType("torchvision")
    tv._ymi_stand_in = True
    tv.__version__ = "0.0.0+standin"
    tv._is_tracing = _is_tracing
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")
    for m in (ops, boxes):
        m.box_convert = box_convert
        m.nms = nms
        m.batched_nms = batched_nms
    def box_iou(a, b):
        area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[:, :2]); rb = torch.min(a[:, None, 2:], b[:, 2:])
        wh = (rb - lt).clamp(min=0); inter = wh[..., 0] * wh[..., 1]
        return inter / (area1[:, None] + area2 - inter)
    ops.box_iou = boxes.box_iou = box_iou
    ops.boxes = boxes
    models = types.ModuleType("torchvision.models")
    mutils = types.ModuleType("torchvision.models._utils")
    mutils.IntermediateLayerGetter = IntermediateLayerGetter
    models._utils = mutils
    io = types.ModuleType("torchvision.io")
    io.read_image = read_image
    io.ImageReadMode = ImageReadMode
    tv.ops, tv.models, tv.io = ops, models, io
    sys.modules.update({
        "torchvision": tv, "torchvision.ops": ops, "torchvision.ops.boxes": boxes,
        "torchvision.models": models, "torchvision.models._utils": mutils, "torchvision.io": io,
    })
