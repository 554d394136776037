"""Op-level Python wrappers over the C ABI (tests, stand-alone API pieces).  No fallbacks."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from . import _lib
from ._lib import YmiError, check
from .engine import Plan, View


def batched_nms(boxes: Tensor, scores: Tensor, labels: Tensor, iou_threshold: float) -> Tensor:
    """Class-aware NMS of one image on the MI355X; returns kept indices (int64) in score-descending
    stable order -- the contract of torchvision.ops.batched_nms as called at
    yolort/models/box_head.py:422."""
    lib = _lib.load(require_gpu=True)
    if not boxes.is_cuda:
        raise YmiError("batched_nms runs on an MI355X only (no CPU fallback)")
    n = int(scores.numel())
    dev = boxes.device
    b = boxes.detach().to(torch.float32).contiguous()
    s = scores.detach().to(torch.float32).contiguous()
    # torchvision accepts arbitrary int64 category ids; the kernel's records carry a 12-bit class field.  Only EQUALITY of ids matters to the suppression, so the ids are
    # renumbered densely (sorted unique -> 0 .. u-1) first; more than 4096 distinct classes in one call are refused instead of aliased
    if n > 0:
        uniq, l = torch.unique(labels.detach().reshape(-1), sorted=True, return_inverse=True)
        if uniq.numel() > 4096:
            raise YmiError(f"batched_nms: {uniq.numel()} distinct category ids in one call (the kernel holds 4096)")
        l = l.to(torch.int32).contiguous()
    else:
        l = labels.detach().to(torch.int32).contiguous()
    keep = torch.empty(max(n, 1), device=dev, dtype=torch.int32)
    count = torch.zeros(1, device=dev, dtype=torch.int32)
    ws = torch.empty(lib.ymi_nms_ws_bytes(n), device=dev, dtype=torch.uint8)
    check(lib.ymi_batched_nms(b.data_ptr(), s.data_ptr(), l.data_ptr(), n, float(iou_threshold), keep.data_ptr(), count.data_ptr(),
                              ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "ymi_batched_nms")
    k = int(count.item())
    return keep[:k].to(torch.int64)


def slab_to_list(boxes: Tensor, scores: Tensor, labels: Tensor, counts: Sequence[int]) -> List[Dict[str, Tensor]]:
    """Fixed (N,K,..) slab -> the reference's List[Dict] with keys in its order (box_head.py:427)."""
    out = []
    for i, k in enumerate(counts):
        out.append({"scores": scores[i, :k], "labels": labels[i, :k], "boxes": boxes[i, :k]})
    return out


def postprocess_logits(head_outputs: Sequence[Tensor], strides: Sequence[float], anchors: Sequence[Sequence[float]], num_classes: int,
                       score_thresh: float, nms_thresh: float, detections_per_img: int, cand_cap: Optional[int] = None) -> List[Dict[str, Tensor]]:
    """Runs the fused post-process on reference-layout head outputs [(N,A,H,W,K)] (A == 3)."""
    _lib.load(require_gpu=True)
    h0 = head_outputs[0]
    if not h0.is_cuda:
        raise YmiError("postprocess runs on an MI355X only (no CPU fallback)")
    n = h0.shape[0]
    k = num_classes + 5
    plan_inputs = []
    for ho in head_outputs:
        if ho.shape[1] != 3 or ho.shape[-1] != k:
            raise YmiError(f"expected (N,3,H,W,{k}) head outputs, got {tuple(ho.shape)}")
        cs = (3 * k + 3) // 4 * 4
        nhwc = torch.zeros(n, ho.shape[2], ho.shape[3], cs, device=ho.device, dtype=torch.float32)
        nhwc[..., : 3 * k] = ho.to(torch.float32).permute(0, 2, 3, 1, 4).reshape(n, ho.shape[2], ho.shape[3], 3 * k)
        plan_inputs.append(nhwc)
    cap = cand_cap or max(4096, 2048 * n)
    flags = 0
    while True:
        plan = Plan(h0.device, torch.float16)
        views = [View(t.view(-1), 0, n, t.shape[1], t.shape[2], 3 * k, t.shape[3]) for t in plan_inputs]
        pb = plan.postprocess(views, strides, anchors, num_classes, score_thresh, nms_thresh, detections_per_img, cap, flags=flags)
        plan.run()
        st = pb.status.cpu().tolist()
        if st[1] == 0:
            break
        if not st[1] & 1:   # YMI_STATUS_PREFIX_SHORT only: the score prefix of a crowded image was too short, take everything
            flags = _lib.POST_EXACT_FULL
            continue
        need = max(st[0], st[3] * n)     # st[3]: largest per-image count when the per-image sort path overflowed
        cap = max(int(need * 1.25) + 1024, 2 * cap)  # nothing is truncated silently: grow and redo
        cap = n * (1 << ((cap + n - 1) // n - 1).bit_length())   # per-image regions are powers of two
    counts = pb.count.cpu().tolist()
    return slab_to_list(pb.boxes, pb.scores, pb.labels, counts)
