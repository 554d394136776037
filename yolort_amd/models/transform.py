"""Letterbox pre-processing and inverse box rescale (reference yolort/models/transform.py).

Host side: the reference's size arithmetic, reproduced operation by operation (fp32
reciprocal-multiply scale, double truncation, banker's-rounded pad split -- SURVEY.md Appendix
C-1/C-2).  Device side: ONE launch (csrc/preproc_pool.hip letterbox_kernel) does the bilinear
gather, fill, dtype cast and CHW -> NHWC4 layout change for the whole batch, instead of one
F.interpolate + one copy_ per image (transform.py:181-194, :317-328).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from .. import _lib
from .._lib import YmiError, check, dtype_code
from ..engine import View
from ..hipmodule import compute_dtype_of, view_to_nchw


class NestedTensor:
    """Batched, padded images + their resized sizes (reference transform.py:14-25).

    `view` is the NHWC(4) device buffer the conv stack consumes; `tensors` materialises the
    reference's NCHW tensor on demand (API parity; not used on the fused path)."""

    def __init__(self, view: View, image_sizes: List[Tuple[int, int]]):
        self.view = view
        self.image_sizes = image_sizes

    @property
    def shape(self) -> Tuple[int, int, int, int]:
        return (self.view.n, 3, self.view.h, self.view.w)

    def nchw(self) -> Tensor:
        v = self.view
        return view_to_nchw(View(v.base, v.off, v.n, v.h, v.w, 3, v.cs))

    @property
    def tensors(self) -> Tensor:
        return self.nchw()

    def __iter__(self):  # allows `tensors, image_sizes = nested`
        yield self.tensors
        yield self.image_sizes


def resized_hw(h: int, w: int, min_size: float, max_size: float) -> Tuple[int, int]:
    """Output size of the reference resize (transform.py:66-83): scale = min(S_min/min, S_max/max)
    where `float / 0-dim fp32 tensor` is reciprocal-then-multiply in fp32; the result is widened to
    double by .item() and `int(in * scale)` truncates in double."""
    f = np.float32
    r_min = f(1.0) / f(min(h, w))
    r_max = f(1.0) / f(max(h, w))
    scale = float(min(f(r_min * f(min_size)), f(r_max * f(max_size))))
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))


def pad_offset(canvas: int, size: int) -> int:
    """top/left padding (transform.py:321-326): Python round-half-even of (canvas-size)/2 - 0.1"""
    return int(round((canvas - size) / 2 - 0.1))


def rescale_params(canvas_hw: Tuple[int, int], original_hw: Tuple[int, int]) -> Tuple[float, float, float]:
    """(gain, pad_x, pad_y) of scale_coords (transform.py:358-359), fp32 like the reference's
    int64-tensor / int arithmetic."""
    f = np.float32
    gain = min(f(canvas_hw[0]) / f(original_hw[0]), f(canvas_hw[1]) / f(original_hw[1]))
    pad_x = f(f(canvas_hw[1]) - f(original_hw[1]) * gain) / f(2)
    pad_y = f(f(canvas_hw[0]) - f(original_hw[0]) * gain) / f(2)
    return float(gain), float(pad_x), float(pad_y)


class YOLOTransform(nn.Module):
    """Same constructor as the reference (transform.py:125-141)."""

    def __init__(self, min_size: int, max_size: int, *, size_divisible: int = 32, fixed_shape: Optional[Tuple[int, int]] = None, fill_color: int = 114) -> None:
        super().__init__()
        self.min_size = min_size
        self.max_size = max_size
        self.size_divisible = size_divisible
        self.fixed_shape = fixed_shape
        self.fill_color = fill_color / 255

    # ---- host geometry -------------------------------------------------------------------
    def canvas_of(self, shapes: Sequence[Tuple[int, int]]) -> Tuple[int, int]:
        """the reference's batch canvas for a list of (h, w): `fixed_shape`, or the maximum resized size rounded up to `size_divisible`
        (transform.py:307-314).  A stream that is SHARDED over ranks keeps the reference's detections only when every rank letterboxes onto the canvas of the
        WHOLE list (the reference pads to the maximum over the whole list, :311): evaluate this on the global list of image sizes -- a host computation over
        integers -- and hand the result to every rank's `YOLOv5.forward(..., canvas=...)` (or let the ranks agree on it: yolort_amd.dist.agree_canvas)."""
        return self.geometry(shapes)[0]

    def geometry(self, shapes: Sequence[Tuple[int, int]], canvas: Optional[Tuple[int, int]] = None) -> Tuple[Tuple[int, int], List[Tuple[int, int]], List[Tuple[int, int]]]:
        """canvas (Hb,Wb), resized sizes and (top,left) pads for a list of (h,w) -- batch_images :297-330.  `canvas`: the canvas of a larger list this one is a
        shard of (see canvas_of); it takes the place of this list's own maximum exactly like the reference's `fixed_shape` (:307-308) would."""
        sizes = [resized_hw(h, w, float(self.min_size), float(self.max_size)) for h, w in shapes]
        if canvas is not None:
            hb, wb = int(canvas[0]), int(canvas[1])
            if any(s[0] > hb or s[1] > wb for s in sizes):
                raise ValueError(f"canvas {(hb, wb)} is smaller than a letterboxed image of this batch ({max(s[0] for s in sizes)} x {max(s[1] for s in sizes)}): it must be the canvas of a list that contains this batch")
        elif self.fixed_shape is not None:
            hb, wb = int(self.fixed_shape[0]), int(self.fixed_shape[1])
        else:
            stride = float(self.size_divisible)
            hb = int(math.ceil(float(max(s[0] for s in sizes)) / stride) * stride)
            wb = int(math.ceil(float(max(s[1] for s in sizes)) / stride) * stride)
        pads = [(pad_offset(hb, s[0]), pad_offset(wb, s[1])) for s in sizes]
        return (hb, wb), sizes, pads

    # ---- device -----------------------------------------------------------------------------
    @staticmethod
    def is_hwc(im: Tensor) -> bool:
        """interleaved uint8 (H, W, 3) as image decoders deliver it (a (3, W, 3) uint8 tensor counts as planar)"""
        return im.dtype == torch.uint8 and im.dim() == 3 and im.shape[-1] == 3 and im.shape[0] != 3

    @staticmethod
    def image_hw(im: Tensor) -> Tuple[int, int]:
        return (int(im.shape[0]), int(im.shape[1])) if YOLOTransform.is_hwc(im) else (int(im.shape[-2]), int(im.shape[-1]))

    def letterbox_into(self, images: Sequence[Tensor], out: View, sizes, pads, stream=None, ptrs: Optional[bytes] = None, uniform_kind: bool = False) -> None:
        """`ptrs` / `uniform_kind`: from the C pass over the image list (YOLOv5.forward_async): the data pointers of contiguous images as packed 64-bit words, and the promise
        that the images share dtype and layout"""
        lib = _lib.load(require_gpu=True)
        n = len(images)
        if not uniform_kind:
            kinds = {(im.dtype, self.is_hwc(im)) for im in images}
            if len(kinds) != 1:
                raise YmiError("all images of a batch must share one dtype and layout")
        hwc = self.is_hwc(images[0])
        imgs = images
        if ptrs is not None:
            ptrs = (C.c_void_p * n).from_buffer_copy(ptrs)
        else:
            if not all(im.is_contiguous() for im in images):   # the copies are torch work: they run on (and their memory belongs to) the stream of the launch below
                with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
                    imgs = [im if im.is_contiguous() else im.contiguous() for im in images]
            ptrs = (C.c_void_p * n)(*[im.data_ptr() for im in imgs])
        flat = []
        for i, im in enumerate(imgs):
            sh = im.shape
            flat += [sh[0], sh[1], sizes[i][0], sizes[i][1], pads[i][0], pads[i][1]] if hwc else [sh[-2], sh[-1], sizes[i][0], sizes[i][1], pads[i][0], pads[i][1]]
        geom = (C.c_int32 * (6 * n))(*flat)
        check(lib.ymi_letterbox(ptrs, geom, n, _lib.YMI_U8_HWC if hwc else dtype_code(imgs[0].dtype), out.ptr, out.h, out.w, out.c, dtype_code(out.dtype),
                                C.c_float(self.fill_color), _lib.stream_ptr(stream)), "ymi_letterbox")

    def forward(self, images: Sequence[Tensor], targets=None, dtype: Optional[torch.dtype] = None, out: Optional[View] = None):
        if targets is not None:
            raise NotImplementedError("target transformation belongs to the training path (out of scope)")
        images = [images[i] for i in range(len(images))]
        for im in images:
            if im.dim() != 3:  # reference :185-189
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], but got '{im.shape}'.")
            if im.shape[0] != 3 and not self.is_hwc(im):
                raise ValueError(f"images are expected to have 3 channels, got {im.shape[0]}")
            if not im.is_cuda:
                raise YmiError("yolort_amd runs on an MI355X only: images must live on 'cuda' (there is no CPU fallback)")
        (hb, wb), sizes, pads = self.geometry([self.image_hw(im) for im in images])
        if out is None:
            dt = dtype or (images[0].dtype if images[0].dtype.is_floating_point else torch.float32)
            t = torch.empty(len(images) * hb * wb * 4, device=images[0].device, dtype=dt)
            out = View(t, 0, len(images), hb, wb, 4, 4)
        if (out.n, out.h, out.w) != (len(images), hb, wb):
            raise YmiError("letterbox output view does not match the batch canvas")
        self.letterbox_into(images, out, sizes, pads)
        return NestedTensor(out, [(int(s[0]), int(s[1])) for s in sizes]), None

    def postprocess(self, result: List[Dict[str, Tensor]], image_shapes, original_image_sizes: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        """API parity (transform.py:332-343); the fused path applies the same affine map inside the
        top-k gather kernel instead."""
        hw = (int(image_shapes[0]), int(image_shapes[1]))
        for i, (pred, o) in enumerate(zip(result, original_image_sizes)):
            result[i]["boxes"] = scale_coords(pred["boxes"], hw, o)
        return result

    def __repr__(self):
        return f"{self.__class__.__name__}(\n    Resize(min_size={self.min_size}, max_size={self.max_size})\n)"


def scale_coords(boxes: Tensor, new_size, original_size: Tuple[int, int]) -> Tensor:
    """(box - pad) / gain, no clipping (reference transform.py:354-367)."""
    gain, pad_x, pad_y = rescale_params((int(new_size[0]), int(new_size[1])), (int(original_size[0]), int(original_size[1])))
    b = boxes.to(torch.float32)
    pad = torch.tensor([pad_x, pad_y, pad_x, pad_y], dtype=torch.float32, device=b.device)
    return (b - pad) / gain
