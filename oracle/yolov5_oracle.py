"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the reference's YOLOv5 inference path.

This file is the ORACLE for the MI355X HIP path.  It is a from-scratch functional restatement (no
nn.Module tree) of what `yolort.models.YOLOv5.forward` computes in eval mode, driven directly by a
reference-format ``state_dict``.  Every function cites the reference lines it follows.  It is
imported only by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg; the
product package ``yolort_amd`` never imports it and has no CPU fallback.

Pinning: ``tests/golden/make_golden.py`` runs the UNMODIFIED reference (read-only import from
/root/reference with oracle/_tv_compat.py standing in for the absent torchvision) and stores its
outputs under tests/golden/; ``tests/test_oracle_golden.py`` checks this restatement against those
vectors (letterbox sizes/pixels, anchors, decode KAT, features, head logits, detections).

PARITY UNPINNED for the class-aware NMS arithmetic only: torchvision (third-party, not vendored in
/root/reference, CI pins 0.10.1-0.14.1) is absent, so `batched_nms` follows the contract fixed in
SURVEY.md Appendix C-4 (stable score-descending sort, strict `>`, per-class exact form, fp32 IoU).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

BN_EPS = 1e-3  # yolort/models/darknetv6.py:110-112, path_aggregation_network.py:161-163

ANCHORS_P5 = [  # yolort/models/yolo.py:94-99
    [10, 13, 16, 30, 33, 23],
    [30, 61, 62, 45, 59, 119],
    [116, 90, 156, 198, 373, 326],
]
ANCHORS_P6 = [  # yolort/models/yolo.py:642-647
    [19, 27, 44, 40, 38, 94],
    [96, 68, 86, 152, 180, 137],
    [140, 301, 303, 264, 238, 542],
    [436, 615, 739, 380, 925, 792],
]


# --------------------------------------------------------------------------------------------
# letterbox (yolort/models/transform.py)
# --------------------------------------------------------------------------------------------
def resized_hw(h: int, w: int, min_size: float, max_size: float) -> Tuple[int, int]:
    """Output size of the aspect-preserving resize.

    transform.py:66-68 computes ``scale = min(S_min / min(h,w), S_max / max(h,w))`` where the
    divisions are ``float / 0-dim float32 Tensor`` = ``reciprocal() * float`` in fp32
    (SURVEY.md Appendix C-1); `.item()` widens to double and F.interpolate
    (recompute_scale_factor=True) truncates ``in * scale`` in double.
    """
    f32 = np.float32
    mn, mx = f32(min(h, w)), f32(max(h, w))
    s = min(f32(f32(1.0) / mn) * f32(min_size), f32(f32(1.0) / mx) * f32(max_size))
    s = float(f32(s))
    return int(math.floor(float(h) * s)), int(math.floor(float(w) * s))


def resize_bilinear(img: Tensor, oh: int, ow: int) -> Tensor:
    """transform.py:76-83: bilinear, align_corners=False, no antialias, scale recomputed as in/out.

    Restated with explicit gather/lerp (ATen upsample_bilinear2d semantics): src = (dst+0.5)*in/out
    - 0.5 clamped at 0; x1 = min(x0+1, in-1); fp32 lerp.
    """
    c, h, w = img.shape
    x = img.to(torch.float32)

    def axis(n_in: int, n_out: int):
        scale = np.float32(n_in) / np.float32(n_out)
        d = torch.arange(n_out, dtype=torch.float32)
        src = (d + 0.5) * float(scale) - 0.5
        src = torch.clamp(src, min=0.0)
        i0 = src.floor().to(torch.int64)
        i0 = torch.clamp(i0, max=n_in - 1)
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        l1 = src - i0.to(torch.float32)
        return i0, i1, l1, 1.0 - l1

    y0, y1, ly1, ly0 = axis(h, oh)
    x0, x1, lx1, lx0 = axis(w, ow)
    top = x[:, y0][:, :, x0] * lx0 + x[:, y0][:, :, x1] * lx1
    bot = x[:, y1][:, :, x0] * lx0 + x[:, y1][:, :, x1] * lx1
    return top * ly0[:, None] + bot * ly1[:, None]


def pad_offsets(canvas: int, size: int) -> int:
    """transform.py:321-326: top/left pad = int(round((canvas - size) / 2 - 0.1)) (banker's round)."""
    return int(round((canvas - size) / 2 - 0.1))


def letterbox(
    images: Sequence[Tensor],
    min_size: int = 640,
    max_size: int = 640,
    size_divisible: int = 32,
    fixed_shape: Optional[Tuple[int, int]] = None,
    fill_color: int = 114,
) -> Tuple[Tensor, List[Tuple[int, int]]]:
    """YOLOTransform.forward (transform.py:143-221) for inference (targets=None)."""
    resized = []
    for img in images:
        if img.dim() != 3:  # transform.py:185-189
            raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], but got '{img.shape}'.")
        h, w = img.shape[-2:]
        oh, ow = resized_hw(h, w, float(min_size), float(max_size))
        resized.append(resize_bilinear(img, oh, ow))
    sizes = [(int(r.shape[1]), int(r.shape[2])) for r in resized]
    if fixed_shape is not None:  # transform.py:307-308
        hb, wb = fixed_shape
    else:  # transform.py:310-314
        stride = float(size_divisible)
        hb = int(math.ceil(float(max(s[0] for s in sizes)) / stride) * stride)
        wb = int(math.ceil(float(max(s[1] for s in sizes)) / stride) * stride)
    out = torch.full((len(resized), 3, hb, wb), fill_color / 255, dtype=torch.float32)
    for i, r in enumerate(resized):
        dh, dw = pad_offsets(hb, r.shape[1]), pad_offsets(wb, r.shape[2])
        out[i, :, dh : dh + r.shape[1], dw : dw + r.shape[2]] = r
    return out, sizes


def scale_coords(boxes: Tensor, new_size: Tuple[int, int], original_size: Tuple[int, int]) -> Tensor:
    """transform.py:354-367 with new_size an int64 tensor and original ints: true-division of int64
    tensors by python ints gives fp32 (torch default dtype); no clipping."""
    f32 = np.float32
    gain = min(f32(new_size[0]) / f32(original_size[0]), f32(new_size[1]) / f32(original_size[1]))
    gain = f32(gain)
    pad0 = f32(f32(f32(new_size[1]) - f32(original_size[1]) * gain) / f32(2))
    pad1 = f32(f32(f32(new_size[0]) - f32(original_size[0]) * gain) / f32(2))
    b = boxes.to(torch.float32).clone()
    b[:, 0] = (b[:, 0] - float(pad0)) / float(gain)
    b[:, 2] = (b[:, 2] - float(pad0)) / float(gain)
    b[:, 1] = (b[:, 1] - float(pad1)) / float(gain)
    b[:, 3] = (b[:, 3] - float(pad1)) / float(gain)
    return b


# --------------------------------------------------------------------------------------------
# conv stack (yolort/v5/models/common.py, darknetv6.py, path_aggregation_network.py)
# --------------------------------------------------------------------------------------------
class _Calib:
    """Optional BN calibration (synthetic-weight recipe): when active, every
    Conv-BN sets running_mean/var to the batch statistics of its own conv output."""

    def __init__(self) -> None:
        self.active = False


CALIB = _Calib()


class _Emulate:
    """Optional low-precision emulation of the HIP engine's storage format: when `dtype` is
    torch.float16 / torch.bfloat16 the oracle folds BatchNorm into the conv weights, rounds the folded
    weights and every activation to that dtype (accumulation stays fp32), i.e. it restates exactly
    what the MI355X path stores in HBM.  Used to separate "inherent fp16/bf16 rounding" from bugs."""

    def __init__(self) -> None:
        self.dtype = None
        # optional predicate on the layer prefix (e.g. "backbone.body.4.m.0.cv2", "head.head.1"): only the layers it accepts store their input / weights in
        # `dtype`, every other layer computes on what it is given in fp32 -- the per-layer error budget of tools/error_budget.py (None: every layer)
        self.select = None
        # optional torch.Generator: every rounded tensor takes one extra ulp-level relative perturbation on a random half of its elements --
        # a different, equally valid rounding history (another summation order / tile).  tests/golden/make_golden.py uses a few of them to
        # measure the tolerance a 16-bit path can be held to on a workload BEFORE the tolerance is written into a GPU test.
        self.jitter = None


EMULATE = _Emulate()


class _Trace:
    """Optional per-layer tap (tests/test_parity_gpu.py): when `hook` is set, every conv of the stack reports
    hook(prefix, x, stride, pad) with its fp32 input BEFORE computing -- the per-layer GPU parity test feeds exactly
    that tensor (rounded to the storage type) to the HIP launch that implements the layer."""

    def __init__(self) -> None:
        self.hook = None


TRACE = _Trace()


def _q(x: Tensor, layer: Optional[str] = None) -> Tensor:
    if EMULATE.dtype is None or (EMULATE.select is not None and layer is not None and not EMULATE.select(layer)):
        return x
    y = x.to(EMULATE.dtype).to(torch.float32)
    if EMULATE.jitter is not None:
        g = EMULATE.jitter
        ulp = 2.0 ** (-11 if EMULATE.dtype == torch.float16 else -8)
        sgn = (torch.rand(y.shape, generator=g) < 0.5).to(torch.float32) * (torch.randint(0, 2, y.shape, generator=g).to(torch.float32) * 2 - 1)
        y = (y * (1 + sgn * ulp)).to(EMULATE.dtype).to(torch.float32)
    return y


class _Version:
    """Activation of the `Conv` blocks of the network being evaluated: "silu" (r4.0 / r6.0, common.py:62-63) or "hardswish" (r3.1, :64-65).  `backbone` sets it from the
    state_dict's layout for the duration of one evaluation."""

    def __init__(self) -> None:
        self.act = "silu"


VERSION = _Version()


def _activation(y: Tensor) -> Tensor:
    return F.silu(y) if VERSION.act == "silu" else F.hardswish(y)


def conv_bn_silu(x: Tensor, sd: Dict[str, Tensor], p: str, stride: int = 1, pad: Optional[int] = None) -> Tensor:
    """common.py:42-70 `Conv`: act(BN(conv2d(x))), bias-free conv, pad=k//2 (autopad :35-39); act = SiLU, or Hardswish in an r3.1 network (VERSION)."""
    w = sd[p + ".conv.weight"]
    k = w.shape[-1]
    if TRACE.hook is not None:
        TRACE.hook(p, x, stride, k // 2 if pad is None else pad)
    if EMULATE.dtype is not None:  # folded-BN, low-precision storage emulation (see _Emulate)
        scale = sd[p + ".bn.weight"] / torch.sqrt(sd[p + ".bn.running_var"] + BN_EPS)
        bias = sd[p + ".bn.bias"] - sd[p + ".bn.running_mean"] * scale
        y = F.conv2d(_q(x, p), _q(w * scale.view(-1, 1, 1, 1), p), bias, stride, k // 2 if pad is None else pad)
        return _activation(y)
    y = F.conv2d(x, w, None, stride, k // 2 if pad is None else pad)
    if CALIB.active:
        sd[p + ".bn.running_mean"] = y.mean(dim=(0, 2, 3))
        sd[p + ".bn.running_var"] = y.var(dim=(0, 2, 3), unbiased=False)
    y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, 0.0, BN_EPS)
    return _activation(y)


def focus(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """common.py:210-240 `Focus`: Conv(4 c1, c2, 3) over cat(x[::2, ::2], x[1::2, ::2], x[::2, 1::2], x[1::2, 1::2]) (rows first: `focus_transform` :237-240)"""
    if TRACE.hook is not None:
        TRACE.hook(p, x, 2, 2)   # the block's own input (the image): what the HIP path's stem launch reads
    y = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    return conv_bn_silu(y, sd, p + ".conv")


def bottleneck_csp(x: Tensor, sd: Dict[str, Tensor], p: str, shortcut: bool) -> Tensor:
    """common.py:119-146 `BottleneckCSP` (r3.1): cv4(LeakyReLU_0.1(BN(cat(cv3(m(cv1 x)), cv2 x)))); cv2 / cv3 are bare 1x1 convolutions, the Bottlenecks as in `c3`."""
    y = conv_bn_silu(x, sd, p + ".cv1")
    for j in range(_count(sd, p + ".m")):
        z = conv_bn_silu(conv_bn_silu(y, sd, f"{p}.m.{j}.cv1"), sd, f"{p}.m.{j}.cv2")
        y = _q(_q(y, f"{p}.m.{j}.cv1") + z, f"{p}.m.{j}.cv2") if (shortcut and EMULATE.dtype is not None) else (y + z if shortcut else z)
    w3, w2 = sd[p + ".cv3.weight"], sd[p + ".cv2.weight"]
    if TRACE.hook is not None:
        TRACE.hook(p + ".cv3", y, 1, 0)
        TRACE.hook(p + ".cv2", x, 1, 0)
    if EMULATE.dtype is not None:   # the shared BatchNorm folded into the two bare convolutions, half each (what the HIP path stores)
        scale = sd[p + ".bn.weight"] / torch.sqrt(sd[p + ".bn.running_var"] + BN_EPS)
        bias = sd[p + ".bn.bias"] - sd[p + ".bn.running_mean"] * scale
        c_ = w3.shape[0]
        t = torch.cat((F.conv2d(_q(y, p + ".cv3"), _q(w3 * scale[:c_].view(-1, 1, 1, 1), p + ".cv3"), bias[:c_]),
                       F.conv2d(_q(x, p + ".cv2"), _q(w2 * scale[c_:].view(-1, 1, 1, 1), p + ".cv2"), bias[c_:])), dim=1)
    else:
        t = torch.cat((F.conv2d(y, w3), F.conv2d(x, w2)), dim=1)
        if CALIB.active:
            sd[p + ".bn.running_mean"] = t.mean(dim=(0, 2, 3))
            sd[p + ".bn.running_var"] = t.var(dim=(0, 2, 3), unbiased=False)
        t = F.batch_norm(t, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, 0.0, BN_EPS)
    return conv_bn_silu(F.leaky_relu(t, 0.1), sd, p + ".cv4")


def _count(sd: Dict[str, Tensor], prefix: str) -> int:
    n = 0
    while f"{prefix}.{n}.cv1.conv.weight" in sd:
        n += 1
    return n


def c3(x: Tensor, sd: Dict[str, Tensor], p: str, shortcut: bool) -> Tensor:
    """common.py:149-173 `C3` and :94-116 `Bottleneck` (e=1.0 so c1==c2 and add==shortcut)."""
    y = conv_bn_silu(x, sd, p + ".cv1")
    for j in range(_count(sd, p + ".m")):
        z = conv_bn_silu(conv_bn_silu(y, sd, f"{p}.m.{j}.cv1"), sd, f"{p}.m.{j}.cv2")
        y = _q(_q(y, f"{p}.m.{j}.cv1") + z, f"{p}.m.{j}.cv2") if (shortcut and EMULATE.dtype is not None) else (y + z if shortcut else z)
    return conv_bn_silu(torch.cat((y, conv_bn_silu(x, sd, p + ".cv2")), dim=1), sd, p + ".cv3")


def spp(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """common.py:176-187 `SPP` with k=(5,9,13) (path_aggregation_network.py:110)."""
    x = conv_bn_silu(x, sd, p + ".cv1")
    pools = [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)]
    return conv_bn_silu(torch.cat([x] + pools, 1), sd, p + ".cv2")


def backbone(x: Tensor, sd: Dict[str, Tensor], p: str = "backbone") -> List[Tensor]:
    """BackboneWithPAN.forward (backbone_utils.py:54-57): body layers 0..8 with taps after 4,6,8
    (darknetv6.py:81-96, backbone_utils.py:107-110) then PathAggregationNetwork.forward
    (path_aggregation_network.py:199-239)."""
    b = p + ".body"
    legacy = f"{b}.0.conv.conv.weight" in sd          # Focus stem: darknetv4.py:86 (r3.1 / r4.0)
    csp = f"{b}.2.cv4.conv.weight" in sd              # BottleneckCSP blocks: r3.1 (darknetv4.py:133-136, path_aggregation_network.py:242-245)
    VERSION.act = "hardswish" if csp else "silu"
    try:
        return _backbone(x, sd, p, legacy, bottleneck_csp if csp else c3)
    finally:
        VERSION.act = "silu"


def _backbone(x: Tensor, sd: Dict[str, Tensor], p: str, legacy: bool, block) -> List[Tensor]:
    b = p + ".body"
    taps = []
    if legacy:   # darknetv4.py:86-100: Focus, 3 x [Conv k3 s2, block], Conv k3 s2, SPP; taps after layers 4, 6, 8 (backbone_utils.py:107-110)
        x = focus(x, sd, b + ".0")
        for i in (1, 3, 5):
            x = conv_bn_silu(x, sd, f"{b}.{i}", stride=2)
            x = block(x, sd, f"{b}.{i + 1}", shortcut=True)
            if i + 1 in (4, 6):
                taps.append(x)
        x = conv_bn_silu(x, sd, f"{b}.7", stride=2)
        taps.append(spp(x, sd, f"{b}.8"))
    else:
        x = conv_bn_silu(x, sd, b + ".0", stride=2, pad=2)  # darknetv6.py:81 Conv(3,c,k=6,s=2,p=2)
        for i in (1, 3, 5, 7):
            x = conv_bn_silu(x, sd, f"{b}.{i}", stride=2)
            x = block(x, sd, f"{b}.{i + 1}", shortcut=True)
            if i + 1 in (4, 6, 8):
                taps.append(x)
    # the neck's blocks below: C3 (r4.0 / r6.0) or BottleneckCSP (r3.1)
    q = p + ".pan"
    feats = list(taps)
    if f"{q}.intermediate_blocks.p6.0.conv.weight" in sd:  # path_aggregation_network.py:10-41
        y = conv_bn_silu(feats[-1], sd, f"{q}.intermediate_blocks.p6.0", stride=2)
        feats.append(block(y, sd, f"{q}.intermediate_blocks.p6.1", shortcut=True))
    nf = len(feats)
    inners: List[Tensor] = []
    last = feats[-1]
    for idx in range(nf - 1):  # path_aggregation_network.py:215-224
        i0 = 3 * idx
        # path_aggregation_network.py:109-114: the r6.0 neck opens with the SPP, the r3.1 / r4.0 necks with a block (their backbones end in the SPP)
        last = spp(last, sd, f"{q}.inner_blocks.0") if (idx == 0 and not legacy) else block(last, sd, f"{q}.inner_blocks.{i0}", shortcut=False)
        last = conv_bn_silu(last, sd, f"{q}.inner_blocks.{i0 + 1}")
        inners.insert(0, last)
        last = F.interpolate(last, scale_factor=2.0, mode="nearest")  # nn.Upsample(scale_factor=2)
        last = torch.cat([last, feats[nf - idx - 2]], dim=1)
    inners.insert(0, last)
    results = []
    last = block(inners[0], sd, f"{q}.layer_blocks.0", shortcut=False)  # :230-231
    results.append(last)
    for idx in range(nf - 1):  # :233-237
        last = conv_bn_silu(last, sd, f"{q}.layer_blocks.{2 * idx + 1}", stride=2)
        last = torch.cat([last, inners[idx + 1]], dim=1)
        last = block(last, sd, f"{q}.layer_blocks.{2 * idx + 2}", shortcut=False)
        results.append(last)
    return results


def head(features: List[Tensor], sd: Dict[str, Tensor], p: str = "head", num_anchors: int = 3) -> List[Tensor]:
    """YOLOHead.forward (box_head.py:68-82): biased 1x1 conv, view (N,A,K,H,W) -> (N,A,H,W,K)."""
    outs = []
    for i, f in enumerate(features):
        if TRACE.hook is not None:
            TRACE.hook(f"{p}.head.{i}", f, 1, 0)
        y = F.conv2d(_q(f, f"{p}.head.{i}"), _q(sd[f"{p}.head.{i}.weight"], f"{p}.head.{i}"), sd[f"{p}.head.{i}.bias"])
        n, _, h, w = y.shape
        outs.append(y.view(n, num_anchors, -1, h, w).permute(0, 1, 3, 4, 2).contiguous())
    return outs


def anchors_for(num_levels: int) -> Tuple[List[int], List[List[float]]]:
    if num_levels == 3:
        return [8, 16, 32], ANCHORS_P5
    return [8, 16, 32, 64], ANCHORS_P6


def decode(head_outputs: List[Tensor], strides: Sequence[int], anchor_grids: Sequence[Sequence[float]]) -> Tensor:
    """_concat_pred_logits (box_head.py:328-348) + decode_single (_utils.py:43-62) with the
    AnchorGenerator grids/shifts (anchor_utils.py:19-60) in closed form:
    xy = (sig*2 - 0.5 + (x,y)) * stride ; wh = (sig*2)^2 * (aw,ah).  Returns (N, sum A*H*W, K)."""
    outs = []
    for ho, s, ag in zip(head_outputs, strides, anchor_grids):
        n, a, h, w, k = ho.shape
        sig = torch.sigmoid(ho.to(torch.float32))
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        grid = torch.stack((gx, gy), 2).view(1, 1, h, w, 2)
        anc = (torch.tensor(ag, dtype=torch.float32).view(-1, 2) / float(s) * float(s)).view(1, a, 1, 1, 2)
        xy = (sig[..., 0:2] * 2.0 - 0.5 + grid) * float(s)
        wh = (sig[..., 2:4] * 2.0) ** 2 * anc
        outs.append(torch.cat((xy, wh, sig[..., 4:]), dim=-1).view(n, -1, k))
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------------------------
# class-aware NMS (torchvision.ops.batched_nms contract)
# --------------------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_NMS_LIB = None


def _nms_lib():
    """Plain-C restatement (oracle/nms_ref.c), built by oracle/build.sh / __graft_entry__.build()."""
    global _NMS_LIB
    if _NMS_LIB is None:
        path = os.path.join(_HERE, "libnms_ref.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.ymi_ref_batched_nms.restype = ctypes.c_int
            lib.ymi_ref_batched_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
            _NMS_LIB = lib
        else:
            _NMS_LIB = False
    return _NMS_LIB


def batched_nms_numpy(boxes: np.ndarray, scores: np.ndarray, labels: np.ndarray, thr: float) -> np.ndarray:
    """Pure-numpy statement of the contract (small cases; cross-checks the C version)."""
    n = len(scores)
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    b = boxes.astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(n, bool)
    keep = []
    for ii in range(n):
        i = order[ii]
        if sup[i]:
            continue
        keep.append(i)
        rest = order[ii + 1 :]
        rest = rest[labels[rest] == labels[i]]
        if len(rest) == 0:
            continue
        w = np.maximum(np.float32(0), np.minimum(b[i, 2], b[rest, 2]) - np.maximum(b[i, 0], b[rest, 0]))
        h = np.maximum(np.float32(0), np.minimum(b[i, 3], b[rest, 3]) - np.maximum(b[i, 1], b[rest, 1]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[rest] - inter)
        sup[rest[iou > np.float32(thr)]] = True
    return np.asarray(keep, np.int64)


def batched_nms(boxes: Tensor, scores: Tensor, labels: Tensor, thr: float) -> Tensor:
    """Kept indices in stable score-descending order (box_head.py:422 call site)."""
    n = int(scores.numel())
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = np.ascontiguousarray(boxes.detach().cpu().to(torch.float32).numpy())
    s = np.ascontiguousarray(scores.detach().cpu().to(torch.float32).numpy())
    l = np.ascontiguousarray(labels.detach().cpu().to(torch.int64).numpy())
    lib = _nms_lib()
    if lib:
        keep = np.empty(n, np.int64)
        k = lib.ymi_ref_batched_nms(b.ctypes.data, s.ctypes.data, l.ctypes.data, n, ctypes.c_float(thr), keep.ctypes.data)
        return torch.from_numpy(keep[:k].copy())
    return torch.from_numpy(batched_nms_numpy(b, s, l, thr))


def postprocess(pred: Tensor, score_thresh: float, nms_thresh: float, detections_per_img: int) -> List[Dict[str, Tensor]]:
    """PostProcess.forward per-image loop (box_head.py:414-427) + _decode_pred_logits (:351-360)."""
    dets = []
    for i in range(pred.shape[0]):
        p = pred[i]
        scores = p[:, 5:] * p[:, 4:5]
        cx, cy, w, h = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
        boxes = torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)
        inds, labels = torch.where(scores > score_thresh)
        b, s = boxes[inds], scores[inds, labels]
        keep = batched_nms(b, s, labels, nms_thresh)[:detections_per_img]
        dets.append({"scores": s[keep], "labels": labels[keep], "boxes": b[keep]})
    return dets


# --------------------------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------------------------
def num_levels(sd: Dict[str, Tensor], p: str = "") -> int:
    n = 0
    while f"{p}head.head.{n}.weight" in sd:
        n += 1
    return n


def yolo_forward(x: Tensor, sd: Dict[str, Tensor], score_thresh=0.005, nms_thresh=0.45, detections_per_img=300,
                 p: str = "", return_stages: bool = False):
    """YOLO.forward eval branch (yolo.py:141-183) on an already-batched (N,3,H,W) tensor."""
    feats = backbone(x.to(torch.float32), sd, p + "backbone")
    nl = num_levels(sd, p)
    strides, anchors = anchors_for(nl)
    ho = head(feats, sd, p + "head")
    pred = decode(ho, strides, anchors)
    dets = postprocess(pred, score_thresh, nms_thresh, detections_per_img)
    if return_stages:
        return dets, {"features": feats, "head": ho, "pred": pred}
    return dets


def yolov5_forward(images: Sequence[Tensor], sd: Dict[str, Tensor], size=(640, 640), size_divisible=32, fixed_shape=None,
                   fill_color=114, score_thresh=0.005, nms_thresh=0.45, detections_per_img=300, return_stages=False):
    """YOLOv5.forward eval branch (yolov5.py:135-189); `sd` has the `model.`-prefixed keys."""
    orig = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
    batch, sizes = letterbox(images, size[0], size[1], size_divisible, fixed_shape, fill_color)
    res = yolo_forward(batch, sd, score_thresh, nms_thresh, detections_per_img, p="model.", return_stages=return_stages)
    dets, stages = res if return_stages else (res, None)
    hb, wb = int(batch.shape[-2]), int(batch.shape[-1])
    for d, o in zip(dets, orig):
        d["boxes"] = scale_coords(d["boxes"], (hb, wb), o)
    if return_stages:
        stages["batch"] = batch
        stages["image_sizes"] = sizes
        return dets, stages
    return dets
