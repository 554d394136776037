"""Multi-GPU data parallelism for the inference path: one process per GPU, the image stream is
sharded contiguously across ranks (images are independent: reference box_head.py:414 loops per
image, BatchNorm is in eval mode), weights are replicated, and the ONLY exchange is one all-gather
of the fixed-shape detection slab per batch (RCCL over xGMI when the backend is "nccl").

Replaces the reference's only collective pattern, the pickle + two all_gathers of
yolort/data/distributed.py:6-49, with a fixed-shape wire format: the slab is
(N/G, K, 4) fp32 boxes + (N/G, K) fp32 scores + (N/G, K) int64 labels + (N/G) int32 counts, packed
into one fp32 buffer so a batch costs a single collective (latency-bound: ~0.3 MB per rank).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous split: rank r gets images [r*n/G, (r+1)*n/G) (remainder spread over the first ranks)"""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def agree_canvas(transform, local_shapes, group=None) -> Tuple[int, int]:
    """The reference's batch canvas of the GLOBAL list when every rank only knows the (h, w) of its own shard: each rank computes the canvas of its
    images (`YOLOTransform.canvas_of`: resize rule, maximum, round up to `size_divisible` -- all integers), one MAX all-reduce of two integers combines them.
    max over ranks of ceil(max_r / d) * d == ceil(max over everything / d) * d, so the result is exactly `transform.canvas_of(all shapes)` -- the canvas the
    reference would pad the whole list to (transform.py:307-314).  Host-side and tiny; ranks that already know every image size (the process that shards the
    stream usually does) call `transform.canvas_of(all_shapes)` instead and skip the collective.  An empty shard contributes (0, 0)."""
    hb, wb = transform.canvas_of(local_shapes) if len(local_shapes) else (0, 0)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return hb, wb
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([hb, wb], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hb, wb = (int(v) for v in t.tolist())
    return hb, wb


SLAB_STALE = -1   # count column of a shard whose rank has to re-run the batch locally (see resolve_stale)


def pack_slab(boxes: Tensor, scores: Tensor, labels: Tensor, count: Tensor, stale: Optional[Tensor] = None) -> Tensor:
    """(n,K,4)+(n,K)+(n,K)+(n) -> one fp32 buffer (n, K*6 + 1); labels/counts travel exactly (ints < 2^24).
    HOST-SIDE form (second round of `resolve_stale`, tests): on the serving path the slab is written by the top-k kernel itself
    (include/yolort_amd.h ymi_post_desc.out_slab, round 4) and the collective is enqueued straight behind the post-process.
    `stale`: optional 0-d / 1-element tensor on the same device (no host sync); when non-zero the shard's count column is
    SLAB_STALE -- the sender will re-run this batch (candidate capacity / score-prefix redo) and every rank learns it from the
    collective itself."""
    n, k = scores.shape
    buf = torch.empty(n, k * 6 + 1, device=scores.device, dtype=torch.float32)
    buf[:, : 4 * k] = boxes.reshape(n, 4 * k)
    buf[:, 4 * k: 5 * k] = scores
    buf[:, 5 * k: 6 * k] = labels.to(torch.float32)
    cnt = count.to(torch.float32)
    if stale is not None:
        cnt = torch.where(stale.reshape(-1)[:1] != 0, torch.full_like(cnt, float(SLAB_STALE)), cnt)
    buf[:, 6 * k] = cnt
    return buf


def dets_to_slab(dets: List[Dict[str, Tensor]], k: int, device=None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """List[Dict] (the model's output) -> the fixed-shape slab (n,K,4) fp32, (n,K) fp32, (n,K) int64, (n) int32"""
    dev = device if device is not None else (dets[0]["scores"].device if dets else torch.device("cpu"))
    n = len(dets)
    boxes = torch.zeros(n, k, 4, device=dev)
    scores = torch.zeros(n, k, device=dev)
    labels = torch.zeros(n, k, dtype=torch.int64, device=dev)
    count = torch.zeros(n, dtype=torch.int32, device=dev)
    for i, d in enumerate(dets):
        m = min(k, d["scores"].shape[0])
        boxes[i, :m], scores[i, :m], labels[i, :m], count[i] = d["boxes"][:m].float(), d["scores"][:m].float(), d["labels"][:m], m
    return boxes, scores, labels, count


def resolve_stale(gathered: Tuple[Tensor, Tensor, Tensor, Tensor], local_final: Callable[[], Tuple[Tensor, Tensor, Tensor, Tensor]],
                  group=None) -> Tuple[Tuple[Tensor, Tensor, Tensor, Tensor], bool]:
    """Second round of the per-batch exchange, taken by EVERY rank iff any shard of the first round is marked SLAB_STALE.
    All ranks hold the same gathered counts, so they agree on whether to enter without talking to each other; the ranks
    that re-ran the batch contribute their final results, the others re-send theirs.  Returns (global slab, second_round)."""
    counts = gathered[3]
    if not bool((counts < 0).any().item()):
        return gathered, False
    b, s, l, c = local_final()
    return all_gather_slab(b, s, l, c, group), True


def unpack_slab(buf: Tensor, k: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    n = buf.shape[0]
    return (buf[:, : 4 * k].reshape(n, k, 4), buf[:, 4 * k: 5 * k], buf[:, 5 * k: 6 * k].to(torch.int64), buf[:, 6 * k].to(torch.int32))


def all_gather_slab(boxes: Tensor, scores: Tensor, labels: Tensor, count: Tensor, group=None, global_n: Optional[int] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """One collective per batch; every rank ends up with the detections of the global batch in rank order.
    Equal shard sizes (weak scaling / N divisible by G) need nothing else.  For a `shard_range` split of `global_n` images that leaves a
    remainder, pass `global_n`: every rank pads its slab to ceil(N / G) rows (no extra exchange: all ranks know N and G), the collective
    stays a single fixed-shape all-gather, and the padding rows are dropped afterwards."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return boxes, scores, labels, count
    world = dist.get_world_size(group)
    local = pack_slab(boxes, scores, labels, count)
    rows = local.shape[0]
    if global_n is not None:
        rank = dist.get_rank(group)
        lo, hi = shard_range(global_n, rank, world)
        if hi - lo != rows:
            raise ValueError(f"rank {rank} holds {rows} images, shard_range({global_n}, {rank}, {world}) says {hi - lo}")
        rows = -(-global_n // world)
        if rows != local.shape[0]:
            local = torch.cat([local, local.new_zeros(rows - local.shape[0], local.shape[1])])
    out = torch.empty(world * rows, local.shape[1], device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, local, group=group)
    if global_n is not None and rows * world != global_n:
        keep = torch.cat([torch.arange(r * rows, r * rows + (shard_range(global_n, r, world)[1] - shard_range(global_n, r, world)[0])) for r in range(world)]).to(out.device)
        out = out.index_select(0, keep)
    return unpack_slab(out, scores.shape[1])


def gather_detections(dets: List[Dict[str, Tensor]], k: int, group=None) -> List[Dict[str, Tensor]]:
    """List[Dict] convenience form: pads each image's detections to K, all-gathers, slices back."""
    boxes, scores, labels, count = dets_to_slab(dets, k)
    b, s, l, c = all_gather_slab(boxes, scores, labels, count, group)
    cl = c.cpu().tolist()
    return [{"scores": s[i, : cl[i]], "labels": l[i, : cl[i]], "boxes": b[i, : cl[i]]} for i in range(len(cl))]
