"""`YOLOv5`: letterbox + YOLO + box rescale behind the reference's predict API
(reference yolort/models/yolov5.py:19-297)."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import hipmodule
from .._lib import YmiError
from ..utils import contains_any_tensor
from . import yolo
from .transform import YOLOTransform, rescale_params
from .yolo import YOLO

__all__ = ["YOLOv5"]


class YOLOv5(nn.Module):
    def __init__(
        self,
        arch: Optional[str] = None,
        model: Optional[nn.Module] = None,
        num_classes: int = 80,
        pretrained: bool = False,
        progress: bool = True,
        size: Tuple[int, int] = (640, 640),
        size_divisible: int = 32,
        fixed_shape: Optional[Tuple[int, int]] = None,
        fill_color: int = 114,
        **kwargs: Any,
    ) -> None:
        super().__init__()
        self.arch = arch
        self.num_classes = num_classes
        if model is None:
            model = yolo.__dict__[arch](pretrained=pretrained, progress=progress, num_classes=num_classes, **kwargs)
        self.model = model
        self.transform = YOLOTransform(size[0], size[1], size_divisible=size_divisible, fixed_shape=fixed_shape, fill_color=fill_color)
        self._has_warned = False

    def freeze_weights(self, frozen: bool = True) -> "YOLOv5":
        """see YOLO.freeze_weights (serving mode: the plan key stops re-reading every tensor's version per batch)"""
        self.model.freeze_weights(frozen)
        return self

    def set_compute_dtype(self, dtype: torch.dtype) -> "YOLOv5":
        """see YOLO.set_compute_dtype (torch.float32 = fp32 parity mode for fp32-parameter models)"""
        self.model.set_compute_dtype(dtype)
        return self

    def forward(self, inputs: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None, canvas: Optional[Tuple[int, int]] = None):
        """inputs: iterable of (3,H,W) tensors in 0-1 range (or uint8 0-255), possibly of different sizes
        (reference yolov5.py:135-189).  Returns List[Dict] with boxes in ORIGINAL image coordinates.
        `canvas` (Hb, Wb): letterbox onto the canvas of a LARGER list this batch is a shard of (`transform.canvas_of(all sizes)` /
        `yolort_amd.dist.agree_canvas`): the reference pads to the maximum over the whole list it is given (transform.py:307-314), so a
        dynamic-shape stream sharded over ranks reproduces its single-process detections only with the global canvas."""
        return self.forward_async(inputs, targets, canvas=canvas).result()

    def forward_async(self, inputs: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None, canvas: Optional[Tuple[int, int]] = None):
        """Enqueues one batch and returns a handle whose `.result()` yields forward()'s List[Dict].
        Serving loops keep two batches in flight (submit i+1, then collect i): the host work and the
        sort/NMS tail of batch i then overlap the convolutions of batch i+1."""
        if self.training:
            raise NotImplementedError("yolort_amd implements the inference path only; call .eval() (training is out of scope)")
        if targets is not None:
            raise NotImplementedError("targets belong to the training path (out of scope)")
        images = [inputs[i] for i in range(len(inputs))]
        # the per-image checks (3-d, on the GPU, shape, dtype, contiguity, alignment, data_ptr) in ONE pass in C when yolort_amd/lib/_ymi_sig.so is built (torch_ext/sig_ext.cpp):
        # (all_3d, device | -1, same_dtype, uniform_shape | None, shapes | None, all_contiguous, all_aligned, data_ptrs).  Lists it cannot vouch for take the per-image path below,
        # which raises the reference's messages.
        scan = None
        if hipmodule._SIG_EXT is not None and images:
            try:
                scan = hipmodule._SIG_EXT.images(images)
            except TypeError:
                scan = None
            if scan is not None and not (scan[0] and scan[1] >= 0 and scan[2] and (scan[3] is not None or images[0].dtype != torch.uint8)):
                scan = None   # (uint8 images of different shapes may mix planar and interleaved layouts: checked one by one)
        if scan is not None:
            if scan[3] is not None:
                original = [self.transform.image_hw(images[0])] * len(images)
            else:
                original = [(s_[1], s_[2]) for s_ in scan[4]]
        else:
            for im in images:
                if im.dim() != 3:  # reference transform.py:185-189
                    raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], but got '{im.shape}'.")
                if not im.is_cuda:
                    raise YmiError("yolort_amd runs on an MI355X only: move the model and inputs to 'cuda' (there is no CPU fallback)")
            original = [self.transform.image_hw(im) for im in images]   # (3, H, W) planar, or (H, W, 3) interleaved uint8
        # host geometry (resize / pad / rescale rows) depends on the list of image sizes only: memoised, a serving loop sees
        # the same few size lists again and again
        gkey = (tuple(original), self.transform.min_size, self.transform.max_size, self.transform.size_divisible, self.transform.fixed_shape,
                None if canvas is None else (int(canvas[0]), int(canvas[1])))
        geo = self._geo_cache.get(gkey) if hasattr(self, "_geo_cache") else None
        if geo is None:
            (hb, wb), sizes, pads = self.transform.geometry(original, canvas)
            geo = ((hb, wb), sizes, pads, [rescale_params((hb, wb), o) for o in original],
                   all(o == (hb, wb) for o in original) and all(s_ == (hb, wb) for s_ in sizes))
            if not hasattr(self, "_geo_cache"):
                self._geo_cache = {}
            if len(self._geo_cache) > 256:
                self._geo_cache.clear()
            self._geo_cache[gkey] = geo
        (hb, wb), sizes, pads, rows_cached, identity = geo
        model = self.model
        if not isinstance(model, YOLO):
            raise YmiError("YOLOv5.model must be a yolort_amd YOLO")
        dev = images[0].device
        if dev.index == torch.cuda.current_device():
            return self._enqueue(model, images, (hb, wb), sizes, pads, rows_cached, identity, original, scan)
        with torch.cuda.device(dev):   # plans, streams and launches belong to the images' device
            return self._enqueue(model, images, (hb, wb), sizes, pads, rows_cached, identity, original, scan)

    def _enqueue(self, model: YOLO, images, canvas, sizes, pads, rows_cached, identity, original, scan=None):
        hb, wb = canvas
        e = model._acquire(len(images), hb, wb, images[0].device)
        main = e.main_stream   # every launch below names its stream: no stream context is entered on the default path
        # fixed-size stream: every image already is the canvas (resize = identity, no padding) in the compute dtype ->
        # the stem reads the planar images itself, the letterbox pass and its NHWC4 copy are skipped (bit-identical)
        planar = model.stem_from_planar and identity and e.plan.stem_planar_ok(images, (hb, wb), scan)
        ev0 = None
        first_op = 0
        ptrs = None
        if scan is not None:   # allocated on the caller's stream, read on the instance's
            hipmodule._SIG_EXT.record_stream(images, main.stream_id, main.device_index, main.device_type)
            ptrs = scan[7] if scan[5] else None
        else:
            for im in images:
                im.record_stream(main)
        if planar:
            if model.bracket is not None:   # measurement hook: the conv bracket starts with the stem
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(main)
            first_op = e.plan.stem_from_planar(images, stream=main, ptrs=ptrs)   # 1, or 2 when the stem and body.1 ran as one launch
        elif model.bracket is not None:
            l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l0.record(main)
            self.transform.letterbox_into(images, e.x, sizes, pads, stream=main, ptrs=ptrs, uniform_kind=scan is not None)
            l1.record(main)
            model.bracket["pre"][0].append(l0)
            model.bracket["pre"][1].append(l1)
        else:
            self.transform.letterbox_into(images, e.x, sizes, pads, stream=main, ptrs=ptrs, uniform_kind=scan is not None)
        rows = rows_cached
        if e.post is None:  # custom post_process hook (torch modules: they run in the instance's stream context); rescale afterwards like the reference (yolov5.py:181)
            with torch.cuda.stream(main):
                pend = model._submit_entry(e, None, first_op, ev0, planar=images if planar else None)
                pend.hook_result = self.transform.postprocess(pend.hook_result, (hb, wb), original)
            return pend
        return model._submit_entry(e, rows, first_op, ev0, planar=images if planar else None, main=main)

    @torch.no_grad()
    def predict(self, x: Any, image_loader: Optional[Callable] = None, canvas: Optional[Tuple[int, int]] = None) -> List[Dict[str, Tensor]]:
        """Reference yolov5.py:202-216 (`canvas`: see forward -- the global canvas of a sharded list)."""
        image_loader = image_loader or self.default_loader
        images = self.collate_images(x, image_loader)
        return self.forward(images, canvas=canvas)

    def default_loader(self, img_path: str) -> Tensor:
        """RGB uint8 image as decoded, (H, W, 3); the permute and the /255 of the reference (yolov5.py:218-228) are fused into
        the letterbox kernel's interleaved-uint8 path (decode stays on the host, PIL)."""
        import numpy as np
        from PIL import Image

        arr = np.asarray(Image.open(img_path).convert("RGB"))
        return torch.from_numpy(arr.copy())   # (H, W, 3) uint8: the HWC -> CHW permute and the /255 happen inside the letterbox kernel

    def collate_images(self, samples: Any, image_loader: Callable) -> List[Tensor]:
        """Reference yolov5.py:230-262: everything moves to the model's device; floating inputs take the
        model dtype, uint8 images stay uint8 (the kernel normalises them)."""
        p = next(self.parameters())

        def prep(t: Tensor) -> Tensor:
            t = t.to(p.device)
            return t if t.dtype == torch.uint8 else t.type_as(p)

        if isinstance(samples, Tensor):
            return [prep(samples)]
        if contains_any_tensor(samples):
            return [prep(s) for s in samples]
        if isinstance(samples, str):
            samples = [samples]
        if isinstance(samples, (list, tuple)) and all(isinstance(s, str) for s in samples):
            return [prep(image_loader(s)) for s in samples]
        raise NotImplementedError(
            f"The type of the sample is {type(samples)}, we currently don't support it now, the "
            "samples should be either a tensor, list of tensors, a image path or list of image paths."
        )

    @classmethod
    def load_from_yolov5(cls, checkpoint_path: str, *, size: Tuple[int, int] = (640, 640), size_divisible: int = 32,
                         fixed_shape: Optional[Tuple[int, int]] = None, fill_color: int = 114, **kwargs: Any):
        model = YOLO.load_from_yolov5(checkpoint_path, **kwargs)
        return cls(model=model, size=size, size_divisible=size_divisible, fixed_shape=fixed_shape, fill_color=fill_color)
