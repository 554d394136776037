"""YOLO core model and the architecture table (reference yolort/models/yolo.py).

`YOLO.forward` has the reference's signature and return type (List[Dict] with keys scores / labels
/ boxes, yolo.py:141-183) but executes as one recorded HIP plan:
  NHWC4 batch -> CSPDarknet + PAN (fused conv kernels) -> head (fp32 logits) -> decode + NMS slab.
Constructor injection points (yolo.py:65-81) are honoured: custom `head`, `anchor_generator` or
`post_process` modules are called with the reference's tensor conventions on top of the HIP backbone.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from .._lib import POST_EXACT_FULL, YmiError, check
from ..engine import Plan, View
from ..hipmodule import compute_dtype_of, nchw_to_view, view_to_nchw, weights_signature
from ..ops import slab_to_list
from .anchor_utils import AnchorGenerator
from .backbone_utils import darknet_pan_backbone
from .box_head import PostProcess, YOLOHead

__all__ = [
    "YOLO",
    "yolov5_darknet_pan_n_r60", "yolov5_darknet_pan_n6_r60", "yolov5_darknet_pan_s_r60", "yolov5_darknet_pan_s6_r60",
    "yolov5_darknet_pan_m_r60", "yolov5_darknet_pan_m6_r60", "yolov5_darknet_pan_l_r60", "yolov5_darknet_pan_l6_r60",
    "yolov5_darknet_pan_x_r60", "yolov5_darknet_pan_x6_r60",
    "yolov5_darknet_pan_s_r31", "yolov5_darknet_pan_m_r31", "yolov5_darknet_pan_l_r31",
    "yolov5_darknet_pan_s_r40", "yolov5_darknet_pan_m_r40", "yolov5_darknet_pan_l_r40", "yolov5_darknet_tan_s_r40",
]

_SKIP_POST = os.environ.get("YOLORT_AMD_DEBUG_SKIP_POST", "0") == "1"   # tuning aid (read at import): submit without the post-process

DEFAULT_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]  # yolo.py:94-99
P6_ANCHORS = [[19, 27, 44, 40, 38, 94], [96, 68, 86, 152, 180, 137], [140, 301, 303, 264, 238, 542], [436, 615, 739, 380, 925, 792]]  # yolo.py:642-647


class _PlanDone:
    """The plan's own completion event (csrc/api.cpp ymi_plan_submit) behind the two calls the ring and PendingDetections make on a torch.cuda.Event.  One per plan
    instance: an instance is never resubmitted while the handle of its previous batch is uncollected (YOLO._entry), so the event always belongs to the handle that reads it."""
    __slots__ = ("lib", "handle")

    def __init__(self, plan: Plan):
        self.lib, self.handle = plan.lib, plan.handle

    def query(self) -> bool:
        rc = self.lib.ymi_plan_done_query(self.handle)
        if rc < 0:
            check(rc, "ymi_plan_done_query")
        return rc == 1

    def synchronize(self) -> None:
        check(self.lib.ymi_plan_done_sync(self.handle), "ymi_plan_done_sync")


class _PlanEntry:
    def __init__(self, plan: Plan, x: View, feats: List[View], logits: Optional[List[View]], post, rescale: Optional[Tensor], n_backbone_ops: int):
        self.plan, self.x, self.feats, self.logits, self.post, self.rescale = plan, x, feats, logits, post, rescale
        self.n_backbone_ops = n_backbone_ops
        self.n_conv_ops = plan.num_ops - (1 if post is not None else 0)
        n = x.n
        # pinned staging for the tiny per-batch host<->device traffic (rescale rows in, status+counts out)
        self.rescale_host = torch.zeros(n, 3, dtype=torch.float32).pin_memory() if post is not None else None
        self.result_host = torch.zeros(8 + n, dtype=torch.int32).pin_memory() if post is not None else None
        self.rescale_set = False
        self.rescale_rows = None   # the rows last uploaded to `rescale` (a serving loop sends the same geometry again and again: uploaded once)
        self.done = None        # event recorded after the post-process + result copy of the last submit
        self.outstanding = None  # the PendingDetections of the last submit while the host has not collected it yet
        # stream priorities (tuning aid, default 0 / 0): YOLORT_AMD_POST_PRIORITY / YOLORT_AMD_CONV_PRIORITY, -1 = high
        pp, cp = int(os.environ.get("YOLORT_AMD_POST_PRIORITY", "0")), int(os.environ.get("YOLORT_AMD_CONV_PRIORITY", "0"))
        self.post_stream = torch.cuda.Stream(device=x.base.device, priority=pp) if post is not None else None
        # each plan instance runs its conv stack on its own stream: consecutive batches overlap on the GPU
        # (the tail of one batch's kernels is filled by the next batch's), measured +12 % on yolov5s bs 32
        self.main_stream = torch.cuda.Stream(device=x.base.device, priority=cp)
        # the default serving path submits a batch in two C calls (ymi_plan_begin / ymi_plan_submit: stream dependencies, graph replay, post-process launches, the result
        # copy and the completion event without a torch stream / event object in between); YOLORT_AMD_C_SUBMIT=0 keeps the torch-level sequence (the A/B partner, and the
        # path of measurement brackets and the distributed gather)
        self.c_submit = post is not None and os.environ.get("YOLORT_AMD_C_SUBMIT", "1") != "0"
        self.c_done = _PlanDone(plan) if self.c_submit else None
        self.main_ptr = self.main_stream.cuda_stream
        self.side_ptr = self.post_stream.cuda_stream if post is not None else None
        self.result_bytes = self.result_host.numel() * 4 if post is not None else 0


class PendingDetections:
    """Handle of an in-flight batch (YOLO.submit / YOLOv5.forward_async); `.result()` blocks on its
    completion event only -- later batches keep running on the GPU meanwhile."""

    def __init__(self, owner: "YOLO", entry: _PlanEntry, rows, hook_result=None, planar=None):
        self.owner, self.entry, self.rows, self.hook_result = owner, entry, rows, hook_result
        self.event = entry.done
        self._result: Optional[List[Dict[str, Tensor]]] = None
        self._gathered = None
        self.gather_issued = False   # set by YOLO._submit_entry when this batch's slab all-gather was enqueued
        self.second_round = False    # gathered(): a stale shard made every rank exchange this batch again
        # planar input images when the stem read them directly (entry.x was never filled): a redo must start from them
        self.planar = planar

    def result(self) -> List[Dict[str, Tensor]]:
        """blocks until this batch is done and returns its List[Dict]; idempotent (the first call detaches the results
        from the plan instance's buffers, later calls return the same objects)"""
        if self.hook_result is not None:
            return self.hook_result
        if self._result is None:
            with torch.cuda.device(self.entry.x.base.device):
                self._result = self._collect()
            if self.entry.outstanding is self:
                self.entry.outstanding = None
        return self._result

    def gathered(self):
        """global detection slab (boxes (G*N,K,4), scores, labels, counts) in rank order; needs YOLO.enable_distributed_gather().
        COLLECTIVE in one case: when some rank had to re-run this batch locally (candidate capacity / score-prefix redo) its shard
        of the first exchange is marked stale (dist.SLAB_STALE in the count column, set on the device), every rank sees that in
        its own copy of the slab and all of them take a second all-gather with their final results (dist.resolve_stale) -- so
        call it for every batch, in submission order, on every rank."""
        from .. import dist as ydist
        self.result()
        if self._gathered is None:
            raise YmiError("no global slab for this batch: enable YOLO.enable_distributed_gather() before submitting it")
        seq = getattr(self, "gather_seq", None)
        if seq is not None and isinstance(self._gathered, Tensor):
            done = getattr(self.owner, "_gather_collected", 0)
            if seq != done + 1:   # a skipped batch would leave the ranks in different collectives (a hang, not an error): fail loudly here instead
                raise YmiError(f"gathered() must be called for every batch in submission order on every rank: batch {seq} asked for, batch {done + 1} is next")
            self.owner._gather_collected = seq
        if isinstance(self._gathered, Tensor):
            k = self.entry.post.k
            first = ydist.unpack_slab(self._gathered, k)
            dev = self._gathered.device
            with torch.cuda.device(dev):
                self._gathered, self.second_round = ydist.resolve_stale(first, lambda: ydist.dets_to_slab(self._result, k, dev), self.owner._gather_group)
        return self._gathered

    def _collect(self) -> List[Dict[str, Tensor]]:
        e = self.entry
        self.event.synchronize()
        host = e.result_host.tolist()
        if self.gather_issued:   # detach the global slab too: the instance's buffer is rewritten by its next batch
            self._gathered = e.gathered.clone()   # (this rank's own shard is marked stale when host[1] != 0: see gathered())
        if host[1] != 0 and e.outstanding is self:
            e.outstanding = None   # a redo below may recycle this very instance: it must not wait for this handle again
        if host[1] & 2 and not host[1] & 1:   # the score prefix of a crowded image gave < detections_per_img survivors: exact full pass
            if os.environ.get("YOLORT_AMD_VERBOSE"):
                print("[yolort_amd] score-prefix selection fell short: re-running with the full candidate set", flush=True)
            return self.owner._redo_exact_full(e, self.rows, self.planar)
        if host[1] != 0:   # candidate capacity exceeded (nothing truncated): grow, rebuild, redo synchronously
            n = e.x.n
            per_image = host[3] if host[3] > 0 else (host[0] + n - 1) // n   # status[3]: largest per-image count (per-image sort path)
            if os.environ.get("YOLORT_AMD_VERBOSE"):
                print(f"[yolort_amd] candidate capacity {self.owner.cand_cap_per_image}/image exceeded (status {host[:4]}): growing", flush=True)
            return self.owner._redo_with_capacity(e, self.rows, per_image, self.planar)
        p = e.post
        return slab_to_list(p.boxes.clone(), p.scores.clone(), p.labels.clone(), host[8:])


class YOLO(nn.Module):
    def __init__(
        self,
        backbone: nn.Module,
        num_classes: int,
        strides: Optional[List[int]] = None,
        anchor_grids: Optional[List[List[float]]] = None,
        anchor_generator: Optional[nn.Module] = None,
        head: Optional[nn.Module] = None,
        criterion: Optional[Callable[..., Dict[str, Tensor]]] = None,
        score_thresh: float = 0.005,
        nms_thresh: float = 0.45,
        detections_per_img: int = 300,
        post_process: Optional[nn.Module] = None,
    ):
        super().__init__()
        if not hasattr(backbone, "out_channels"):
            raise ValueError(
                "backbone should contain an attribute out_channels specifying the number of output channels "
                "(assumed to be the same for all the levels)"
            )
        self.backbone = backbone
        strides = [8, 16, 32] if strides is None else strides
        anchor_grids = DEFAULT_ANCHORS if anchor_grids is None else anchor_grids
        self.anchor_generator = anchor_generator if anchor_generator is not None else AnchorGenerator(strides, anchor_grids)
        self.compute_loss = criterion  # training criterion is out of scope; kept only if the caller injects one
        self.num_classes = num_classes
        self.head = head if head is not None else YOLOHead(backbone.out_channels, self.anchor_generator.num_anchors, self.anchor_generator.strides, num_classes)
        self.post_process = post_process if post_process is not None else PostProcess(self.anchor_generator.strides, score_thresh, nms_thresh, detections_per_img)
        self.compute_dtype = torch.float16  # used when parameters are fp32 (see hipmodule.compute_dtype_of)
        # the conv stack of a batch is replayed as ONE captured hipGraph launch (csrc/api.cpp ymi_plan_run: captured once per plan instance and op range) instead of ~45
        # kernel launches from the host: same kernels, same order, same results (tests/test_boundary_gpu.py), 0.12 ms less host time per batch (profiles/r04n_pipeline_depth_graph.txt).
        # Default since round 5; YOLORT_AMD_GRAPH=0 restores the per-kernel launches.
        self.use_graph = os.environ.get("YOLORT_AMD_GRAPH", "1") != "0"
        # round 6: the post-process range (memsets, selection, sort, NMS, top-k: ~14 enqueues) is a captured graph launch of its own on the side stream -- the same
        # launches in the same order; YOLORT_AMD_POST_GRAPH=0 restores the per-kernel enqueues (the A/B partner of tests/test_boundary_gpu.py)
        self.post_graph = os.environ.get("YOLORT_AMD_POST_GRAPH", "1") != "0"
        self.cand_cap_per_image = int(os.environ.get("YOLORT_AMD_CAND_CAP", "16384"))
        self._entries: Dict[Tuple, _PlanEntry] = {}
        self._ring: Dict[Tuple, List[_PlanEntry]] = {}
        # decode + threshold inside the head convolution's epilogue (the fp32 logits never reach memory); False keeps the
        # logits as plan buffers (`entry.logits`) and decodes them in the post-process op -- identical detections
        self.fuse_head_decode = os.environ.get("YOLORT_AMD_FUSED_HEAD", "1") != "0"
        # identity-size batches of compute-dtype planar images feed the stem directly (no letterbox pass); see YOLOv5.forward_async
        self.stem_from_planar = os.environ.get("YOLORT_AMD_STEM_PLANAR", "1") != "0"
        # True once a batch needed the full candidate set (include/yolort_amd.h YMI_POST_EXACT_FULL); sticky for this model
        self.post_exact_full = os.environ.get("YOLORT_AMD_POST_EXACT_FULL", "0") == "1"
        self.pipeline_depth = 4   # plan instances per shape (built lazily): later batches run while batch i is post-processed / collected
        self.max_shapes = int(os.environ.get("YOLORT_AMD_MAX_SHAPES", "4"))                       # LRU of (batch, canvas) shapes with live plans
        self.max_plan_bytes = int(float(os.environ.get("YOLORT_AMD_MAX_PLAN_GB", "96")) * 2**30)    # activation memory bound of that LRU
        self._has_warned = False
        # multi-GPU serving (yolort_amd/dist.py): when set, every batch's detection slab is all-gathered over RCCL from the
        # post-process stream itself -- no host synchronisation between the post-process and the collective
        self._gather_group = None
        self._gather_on = False
        # measurement hook (bench.py): {"pre": ([], []), "conv": ([], []), "post": ([], [])} -> HIP event pairs recorded on
        # the streams the kernels are launched on, around the letterbox launch, the conv launches and the post-process
        self.bracket = None

    # ------------------------------------------------------------------------------------------
    def set_compute_dtype(self, dtype: torch.dtype) -> "YOLO":
        """compute / storage type of the conv stack while the parameters are fp32: torch.float16 (default), torch.bfloat16,
        or torch.float32 = parity mode (exact fp32 arithmetic, csrc/conv_f32.hip).  fp16 / bf16 parameter models always
        compute in their own dtype."""
        if dtype not in (torch.float16, torch.bfloat16, torch.float32):
            raise ValueError(f"unsupported compute dtype {dtype}")
        self.compute_dtype = dtype
        return self

    def enable_distributed_gather(self, group=None, force: bool = False) -> "YOLO":
        """One process per GPU, the image stream sharded across ranks (dist.shard_range): after this call every submitted
        batch also all-gathers its fixed-shape detection slab (one `all_gather_into_tensor` of (N, 6K+1) fp32, RCCL when the
        backend is "nccl"), enqueued behind the post-process on its stream; `PendingDetections.gathered()` returns the global
        slab in rank order.  No-op for world size 1 unless `force` (tests)."""
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self._gather_group, self._gather_on = group, bool(on)
        return self

    _in_redo = False
    _frozen_signature = None

    def freeze_weights(self, frozen: bool = True) -> "YOLO":
        """Serving mode: the caller promises not to modify, re-allocate or replace any parameter / buffer until `freeze_weights(False)` (or another
        `load_state_dict` / `.to()` followed by a fresh `freeze_weights()`).  The plan cache then keys on the signature taken HERE instead of re-reading
        `_version` / `data_ptr()` of every tensor on every submitted batch (0.1 ms of the host's ~0.36 ms per batch on yolov5s; with `YOLORT_AMD_GRAPH=1` the
        submit path is then ~0.15 ms).  Off by default: an in-place weight update between two batches is otherwise always seen."""
        self._frozen_signature = weights_signature(self) if frozen else None
        return self

    def export_plan(self, path: str, n: int, h: int, w: int, device: Optional[torch.device] = None) -> Dict[str, object]:
        """Records (or reuses) the plan of an (n, h, w) canvas and writes it to `path` as a self-contained file: a C++ consumer replays it with ymi_plan_import +
        ymi_plan_run, no Python (include/yolort_amd.h; the in-scope counterpart of the reference's yolort/relay export).  The file holds the
        canvas form of the plan (op 0 reads the letterboxed NHWC4 batch: ymi_letterbox fills it), the post-process included.  Returns a description of the IO regions."""
        from .._lib import TAG_BOXES, TAG_INPUT, TAG_LABELS, TAG_RESCALE, TAG_SCORES, TAG_SLAB, TAG_STATUS_COUNT
        device = device or next(self.parameters()).device
        e = self._entry(n, h, w, device)
        if e.post is None:
            raise YmiError("export_plan needs the fused post-process (head, anchor generator and post-process of this package)")
        io = {TAG_INPUT: e.x.base, TAG_RESCALE: e.rescale, TAG_BOXES: e.post.boxes, TAG_SCORES: e.post.scores, TAG_LABELS: e.post.labels,
              TAG_STATUS_COUNT: e.post.status_count, TAG_SLAB: e.post.slab}
        nreg = e.plan.export(path, io)
        return {"path": path, "regions": nreg, "ops": e.plan.num_ops, "n_conv_ops": e.n_backbone_ops if hasattr(e, "n_backbone_ops") else None,
                "input": {"shape": (n, h, w, 4), "dtype": str(e.x.dtype)}, "detections_per_img": int(e.post.k)}

    def fused(self) -> bool:
        return type(self.head) is YOLOHead and type(self.post_process) is PostProcess and type(self.anchor_generator) is AnchorGenerator and hasattr(self.backbone, "emit")

    def _entry(self, n: int, h: int, w: int, device: torch.device) -> _PlanEntry:
        """a plan instance of this shape whose previous batch has been collected by the host.  Instances are built lazily
        (one at first use, another only when a batch is submitted while every existing one is still in flight, up to
        `pipeline_depth`); when all are busy the oldest is recycled after its uncollected results have been moved into
        their handle (PendingDetections) -- a handle never reads buffers that a later batch has overwritten.  Shapes are
        kept in a small LRU (`max_shapes`, `max_plan_bytes`): variable-size letterbox streams alternate between a few
        canvases without rebuilding plans."""
        cdt = compute_dtype_of(self)
        pp = self.post_process
        post_key = (pp.score_thresh, pp.nms_thresh, pp.detections_per_img) if self.fused() else None

        base = (n, h, w, cdt, device.index, self._frozen_signature if self._frozen_signature is not None else weights_signature(self), post_key)   # (walks every parameter: once per submission, the host side of a step is ~0.3 ms)

        def current_key():
            return base + (self.cand_cap_per_image, self.fuse_head_decode, self.post_exact_full)

        # Collecting an instance's outstanding batch can trigger a redo that GROWS cand_cap_per_image / sets post_exact_full -- both part of the
        # key -- and re-enters this function: a key computed before the collection would hand back a plan of the capacity that has
        # just overflowed.  So: when every instance of the current key is busy and the ring is full, collect the oldest FIRST, then (re)compute
        # the key and pick the instance.
        ring = self._ring.get(current_key())
        if ring is not None and len(ring) >= max(1, self.pipeline_depth) and not any(c.outstanding is None and (c.done is None or c.done.query()) for c in ring):
            oldest = ring[0]
            if oldest.outstanding is not None:
                oldest.outstanding.result()   # host-synchronises on that batch and detaches its results from the instance's buffers (may redo, may change the key)
        key = current_key()
        if len(self._ring) > 1:
            for stale in [k for k in self._ring if k[:7] == base and k != key]:   # rings of this shape built for a superseded capacity / exactness setting
                for en in self._ring.pop(stale, []):
                    if en.outstanding is not None:
                        en.outstanding.result()
            key = current_key()
        ring = self._ring.get(key)
        if ring is None:
            if not hasattr(self.backbone, "emit"):
                raise YmiError("the backbone must be a yolort_amd HIP module (custom torch backbones have no MI355X path)")
            ring = self._ring[key] = []
        else:
            self._ring[key] = self._ring.pop(key)   # most recently used last
        e = None
        for cand in ring:   # a free instance: nothing outstanding and its GPU work has drained
            if cand.outstanding is None and (cand.done is None or cand.done.query()):
                e = cand
                break
        if e is None and len(ring) < max(1, self.pipeline_depth):
            self._evict(key)
            e = self._build_entry(n, h, w, device, cdt, pp)
            ring.append(e)
        if e is None:       # all busy (their batches are collected but the GPU work of a later use has not drained): recycle the oldest
            e = ring.pop(0)
            ring.append(e)
            if e.outstanding is not None:
                e.outstanding.result()
        self._entries = {key: e}
        return e

    def _evict(self, keep_key) -> None:
        """drop least-recently-used shapes beyond `max_shapes` / `max_plan_bytes` (never the shape in use; instances with an
        uncollected batch are collected first)"""
        def total() -> int:
            return sum(en.plan.bytes_allocated for r in self._ring.values() for en in r)
        while len(self._ring) > 1 and (len(self._ring) > self.max_shapes or total() > self.max_plan_bytes):
            victim = next(k for k in self._ring if k != keep_key)
            for en in self._ring.pop(victim):
                if en.outstanding is not None:
                    en.outstanding.result()

    def _build_entry(self, n: int, h: int, w: int, device: torch.device, cdt, pp) -> _PlanEntry:
        plan = Plan(device, cdt)
        x = plan.alloc(n, h, w, 4, zero=True)
        feats = self.backbone.emit(plan, x)
        n_backbone = plan.num_ops
        plan.set_fuse_stem(os.environ.get("YOLORT_AMD_FUSE_STEM_CANVAS", "1") != "0")   # stem + body.1 as one launch on the letterboxed canvas too
        logits = post = rescale = None
        if self.fused():
            rescale = torch.zeros(n, 3, device=device, dtype=torch.float32)
            ag = self.anchor_generator
            strides = [float(s) for s in ag.strides]
            args = (self.num_classes, float(pp.score_thresh), float(pp.nms_thresh), int(pp.detections_per_img), self.cand_cap_per_image * n)
            flags = POST_EXACT_FULL if self.post_exact_full else 0
            if self.fuse_head_decode and self.head.can_fuse_decode(plan, feats):
                post, pd = plan.post_desc([(f.h, f.w) for f in feats], n, strides, ag.anchor_grids, *args, rescale=rescale, flags=flags)
                plan.post_begin(pd)
                self.head.emit_fused(plan, feats, pd)
                plan.post_finish(pd, post.total_anchors)
            else:
                logits = self.head.emit(plan, feats)
                post = plan.postprocess(logits, strides, ag.anchor_grids, *args, rescale=rescale, flags=flags)
        return _PlanEntry(plan, x, feats, logits, post, rescale, n_backbone)

    def _submit_entry(self, e: _PlanEntry, rescale_rows: Optional[List[Tuple[float, float, float]]], first_op: int = 0, ev0=None, planar=None, main=None) -> PendingDetections:
        """input view already filled on the current stream; enqueues the plan and returns a handle.
        Conv stack on the current stream, post-process + result copy on the entry's side stream, so
        the next batch's convolutions overlap this batch's sort/NMS (few, long-running waves)."""
        if e.post is None:  # custom hooks: HIP backbone, then the injected modules on torch tensors
            # from `first_op`: a planar-stem batch has already run ops 0 (and 1) from the images themselves and never filled the NHWC4 canvas
            e.plan.run(first_op, -1, graph=self.use_graph)   # (current stream: the callers enter the instance's stream for this branch)
            feats = [view_to_nchw(v) for v in e.feats]
            head_outputs = self.head(feats)
            grids, shifts = self.anchor_generator(feats)
            return PendingDetections(self, e, None, hook_result=self.post_process(head_outputs, grids, shifts))
        if main is None:
            main = torch.cuda.current_stream()
        if not e.rescale_set or e.rescale_rows is not rescale_rows:   # (the memoised geometry hands over the SAME list object for the same size list)
            if rescale_rows is None:
                e.rescale_host.zero_()
            else:
                e.rescale_host.copy_(torch.tensor(rescale_rows, dtype=torch.float32))
            with torch.cuda.stream(main):
                e.rescale.copy_(e.rescale_host, non_blocking=True)
            e.rescale_rows, e.rescale_set = rescale_rows, True
        br = self.bracket
        gather = self._gather_on and not self._in_redo
        if e.c_submit and br is None and not gather and not _SKIP_POST:
            check(e.plan.lib.ymi_plan_submit(e.plan.handle, first_op, e.n_conv_ops, (3 if self.post_graph else 1) if self.use_graph else 0, main.cuda_stream, e.side_ptr, e.post.status_count.data_ptr(),
                                             e.result_host.data_ptr(), e.result_bytes, 1 if self.pipeline_depth <= 1 else 0), "ymi_plan_submit")
            e.done = e.c_done
            pd = PendingDetections(self, e, rescale_rows, planar=planar)
            e.outstanding = pd
            return pd
        if br is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            if ev0 is None:   # the caller already started the bracket when it issued op 0 itself (stem from planar images)
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(main)
            e.plan.run(first_op, e.n_conv_ops, graph=self.use_graph, stream=main)
            ev1.record(main)
            br["conv"][0].append(ev0)
            br["conv"][1].append(ev1)
        else:
            e.plan.run(first_op, e.n_conv_ops, graph=self.use_graph, stream=main)
        side = e.post_stream
        side.wait_stream(main)
        if not _SKIP_POST:   # tuning aid: upper bound without sort/NMS
            if br is not None:
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record(side)
                e.plan.run(e.n_conv_ops, -1, stream=side)
                p1.record(side)
                br["post"][0].append(p0)
                br["post"][1].append(p1)
            else:
                e.plan.run(e.n_conv_ops, -1, stream=side)
        with torch.cuda.stream(side):
            e.result_host.copy_(e.post.status_count, non_blocking=True)
            if gather:   # the collective waits for the post-process on `side`, and `side` then waits for it (no host sync)
                import torch.distributed as dist

                from .. import dist as ydist
                # the wire slab was written by the top-k kernel itself (ymi_post_desc.out_slab), stale marker included: nothing between the post-process and the collective
                packed = e.post.slab
                world = dist.get_world_size(self._gather_group)
                if getattr(e, "gathered", None) is None or e.gathered.shape[0] != world * packed.shape[0]:
                    e.gathered = torch.empty(world * packed.shape[0], packed.shape[1], device=packed.device, dtype=packed.dtype)
                work = dist.all_gather_into_tensor(e.gathered, packed, group=self._gather_group, async_op=True)
                work.wait()
        e.done = torch.cuda.Event()
        e.done.record(side)
        if self.pipeline_depth <= 1:
            main.wait_event(e.done)
        pd = PendingDetections(self, e, rescale_rows, planar=planar)
        pd.gather_issued = gather
        if gather:   # gathered() may hold a collective (second round): it has to be taken batch by batch, in submission order, on every rank
            pd.gather_seq = self._gather_submitted = getattr(self, "_gather_submitted", 0) + 1
        e.outstanding = pd
        return pd

    def _acquire(self, n: int, h: int, w: int, device: torch.device) -> _PlanEntry:
        """next plan instance of the ring for this shape; waits (on the GPU, not the host) until the
        work previously submitted on it has drained before its buffers are overwritten"""
        e = self._entry(n, h, w, device)
        if e.c_submit:
            if e.done is not None and e.done is not e.c_done:   # the previous batch went the torch-level way (bracket / gather)
                e.main_stream.wait_event(e.done)
            # inputs were produced on the caller's stream; the instance's previous batch must have drained: both dependencies in one call
            check(e.plan.lib.ymi_plan_begin(e.plan.handle, torch.cuda.current_stream().cuda_stream, e.main_ptr), "ymi_plan_begin")
            return e
        e.main_stream.wait_stream(torch.cuda.current_stream())   # inputs were produced on the caller's stream
        if e.done is not None:
            e.main_stream.wait_event(e.done)
        return e

    def _resubmit(self, e_old: _PlanEntry, rescale_rows, planar) -> List[Dict[str, Tensor]]:
        """re-run a batch synchronously on a (re)built plan instance: from the planar images when the stem read those
        directly (the NHWC4 buffer of the old entry was never filled then), else from the old entry's input buffer"""
        x_old = e_old.x
        torch.cuda.synchronize(x_old.base.device)
        e2 = self._entry(x_old.n, x_old.h, x_old.w, x_old.base.device)
        self._in_redo = True   # a local re-run must not issue a collective the other ranks do not (PendingDetections.gathered raises)
        try:
            return self._resubmit_on(e_old, e2, rescale_rows, planar)
        finally:
            self._in_redo = False

    def _resubmit_on(self, e_old: _PlanEntry, e2: _PlanEntry, rescale_rows, planar) -> List[Dict[str, Tensor]]:
        x_old = e_old.x
        with torch.cuda.device(x_old.base.device), torch.cuda.stream(e2.main_stream):
            if planar is not None and e2.plan.stem_planar_ok(planar, (x_old.h, x_old.w)):
                first_op = e2.plan.stem_from_planar(planar)
                return self._submit_entry(e2, rescale_rows, first_op, planar=planar).result()
            if planar is not None:
                raise YmiError("internal: a planar-stem batch cannot be re-run on a plan without the planar stem")
            e2.x.base.copy_(x_old.base)
            return self._submit_entry(e2, rescale_rows).result()

    def _redo_with_capacity(self, e: _PlanEntry, rescale_rows, needed_per_image: int, planar=None) -> List[Dict[str, Tensor]]:
        # per-image regions are powers of two (the kernel rounds the capacity DOWN to one): grow to the next power
        # of two that holds the largest image.  Batches submitted before an earlier growth land here too and
        # simply re-run on the already grown plan.
        cap_eff = 1 << (self.cand_cap_per_image.bit_length() - 1)
        if needed_per_image > cap_eff or e.post.cand_cap >= self.cand_cap_per_image * e.x.n:
            want = max(int(needed_per_image * 1.25) + 1024, 2 * cap_eff)
            self.cand_cap_per_image = 1 << (want - 1).bit_length()
        return self._resubmit(e, rescale_rows, planar)

    def _redo_exact_full(self, e: _PlanEntry, rescale_rows, planar=None) -> List[Dict[str, Tensor]]:
        self.post_exact_full = True
        return self._resubmit(e, rescale_rows, planar)

    def _run_entry(self, e: _PlanEntry, rescale_rows: Optional[List[Tuple[float, float, float]]]) -> List[Dict[str, Tensor]]:
        return self._submit_entry(e, rescale_rows).result()

    def submit(self, samples: Tensor) -> PendingDetections:
        """asynchronous form of forward(): enqueue a pre-batched (N,3,H,W) tensor, collect later"""
        if self.training:
            raise NotImplementedError("yolort_amd implements the inference path only; call .eval() (training / SetCriterion are out of scope)")
        if not isinstance(samples, Tensor) or samples.dim() != 4:
            raise ValueError("samples is expected to be a batched tensor of shape [N, 3, H, W]")
        if not samples.is_cuda:
            raise YmiError("yolort_amd runs on an MI355X only: move the model and inputs to 'cuda' (there is no CPU fallback)")
        n, c, h, w = samples.shape
        with torch.cuda.device(samples.device):   # plans, streams and launches belong to the inputs' device, whatever the caller's current one is
            e = self._acquire(n, h, w, samples.device)
            with torch.cuda.stream(e.main_stream):
                nchw_to_view(e.plan, samples, 4, out=e.x)
                return self._submit_entry(e, None)

    def forward(self, samples: Tensor, targets: Optional[Tensor] = None):
        """samples: batched images (N,3,H,W) in 0-1 range (reference yolo.py:141-183)."""
        if self.training:
            raise NotImplementedError("yolort_amd implements the inference path only; call .eval() (training / SetCriterion are out of scope)")
        if not isinstance(samples, Tensor) or samples.dim() != 4:
            raise ValueError("samples is expected to be a batched tensor of shape [N, 3, H, W]")
        if not samples.is_cuda:
            raise YmiError("yolort_amd runs on an MI355X only: move the model and inputs to 'cuda' (there is no CPU fallback)")
        return self.submit(samples).result()

    @classmethod
    def load_from_yolov5(cls, checkpoint_path: str, score_thresh: float = 0.25, nms_thresh: float = 0.45, version: str = "r6.0",
                         post_process: Optional[nn.Module] = None):
        """Load model state from a checkpoint trained by ultralytics/yolov5 (reference yolo.py:185-223)."""
        from ._checkpoint import load_from_ultralytics

        info = load_from_ultralytics(checkpoint_path, version=version)
        backbone_name = f"darknet_{info['size']}_{version.replace('.', '_')}"
        backbone = darknet_pan_backbone(backbone_name, info["depth_multiple"], info["width_multiple"], version=version, use_p6=info["use_p6"])
        model = cls(backbone, info["num_classes"], strides=info["strides"], anchor_grids=info["anchor_grids"], score_thresh=score_thresh,
                    nms_thresh=nms_thresh, post_process=post_process)
        model.load_state_dict(info["state_dict"])
        return model


def build_model(backbone_name: str, depth_multiple: float, width_multiple: float, version: str, weights_name: Optional[str] = None,
                pretrained: bool = False, progress: bool = True, num_classes: int = 80, use_p6: bool = False, **kwargs: Any) -> YOLO:
    """Reference yolo.py:226-265.  There is no network here, so `pretrained=True` raises instead of downloading."""
    backbone = darknet_pan_backbone(backbone_name, depth_multiple, width_multiple, version=version, use_p6=use_p6)
    model = YOLO(backbone, num_classes, **kwargs)
    if pretrained:
        raise ValueError(f"No checkpoint is available for model {weights_name} (offline build: load a state_dict explicitly)")
    return model


def _r60(size: str, depth: float, width: float, p6: bool):
    def factory(pretrained: bool = False, progress: bool = True, num_classes: int = 80, **kwargs: Any) -> YOLO:
        extra = dict(use_p6=True, strides=[8, 16, 32, 64], anchor_grids=P6_ANCHORS) if p6 else {}
        name = f"{size}6" if p6 else size
        return build_model(f"darknet_{size}_r6_0", depth, width, "r6.0", f"yolov5_darknet_pan_{name}_r60_coco", pretrained=pretrained,
                           progress=progress, num_classes=num_classes, **extra, **kwargs)

    factory.__doc__ = f"yolov5 {size}{'6' if p6 else ''} release 6.0 (depth {depth}, width {width}); reference yolo.py:472-834"
    return factory


yolov5_darknet_pan_n_r60 = _r60("n", 0.33, 0.25, False)
yolov5_darknet_pan_s_r60 = _r60("s", 0.33, 0.5, False)
yolov5_darknet_pan_m_r60 = _r60("m", 0.67, 0.75, False)
yolov5_darknet_pan_l_r60 = _r60("l", 1.0, 1.0, False)
yolov5_darknet_pan_x_r60 = _r60("x", 1.33, 1.25, False)
yolov5_darknet_pan_n6_r60 = _r60("n", 0.33, 0.25, True)
yolov5_darknet_pan_s6_r60 = _r60("s", 0.33, 0.5, True)
yolov5_darknet_pan_m6_r60 = _r60("m", 0.67, 0.75, True)
yolov5_darknet_pan_l6_r60 = _r60("l", 1.0, 1.0, True)
yolov5_darknet_pan_x6_r60 = _r60("x", 1.33, 1.25, True)


def _legacy(size: str, depth: float, width: float, version: str):
    tag = version.replace(".", "").replace("r", "r")   # "r31" / "r40"

    def factory(pretrained: bool = False, progress: bool = True, num_classes: int = 80, **kwargs: Any) -> YOLO:
        return build_model(f"darknet_{size}_{version.replace('.', '_')}", depth, width, version, f"yolov5_darknet_pan_{size}_{tag}_coco", pretrained=pretrained,
                           progress=progress, num_classes=num_classes, **kwargs)

    factory.__doc__ = f"yolov5{size} release {version[1:]} (Focus stem, {'BottleneckCSP / Hardswish' if version == 'r3.1' else 'C3 / SiLU'}; depth {depth}, width {width}); reference yolo.py:292-469"
    return factory


yolov5_darknet_pan_s_r31 = _legacy("s", 0.33, 0.5, "r3.1")
yolov5_darknet_pan_m_r31 = _legacy("m", 0.67, 0.75, "r3.1")
yolov5_darknet_pan_l_r31 = _legacy("l", 1.0, 1.0, "r3.1")
yolov5_darknet_pan_s_r40 = _legacy("s", 0.33, 0.5, "r4.0")
yolov5_darknet_pan_m_r40 = _legacy("m", 0.67, 0.75, "r4.0")
yolov5_darknet_pan_l_r40 = _legacy("l", 1.0, 1.0, "r4.0")


def yolov5_darknet_tan_s_r40(*args: Any, **kwargs: Any):
    raise NotImplementedError("yolov5_darknet_tan_s_r40: the transformer neck (C3TR, reference yolo.py:837-867) is out of the MI355X hot-path scope")
