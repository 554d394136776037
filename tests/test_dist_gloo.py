"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: contiguous sharding and the
single fixed-shape slab all-gather that bench.py runs over RCCL on the GPU box."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yolort_amd import dist as yd
        n, k = 6, 5
        lo, hi = yd.shard_range(n, rank, world)
        g = torch.Generator().manual_seed(1234)
        boxes = torch.rand(n, k, 4, generator=g)
        scores = torch.rand(n, k, generator=g)
        labels = torch.randint(0, 80, (n, k), generator=g)
        count = torch.tensor([5, 0, 3, 1, 5, 2], dtype=torch.int32)
        b, s, l, c = yd.all_gather_slab(boxes[lo:hi], scores[lo:hi], labels[lo:hi], count[lo:hi])
        ok = torch.equal(b, boxes) and torch.equal(s, scores) and torch.equal(l, labels) and torch.equal(c, count)
        dets = [{"boxes": boxes[i, : count[i]], "scores": scores[i, : count[i]], "labels": labels[i, : count[i]]} for i in range(lo, hi)]
        full = yd.gather_detections(dets, k)
        ok = ok and len(full) == n and all(torch.equal(full[i]["labels"], labels[i, : count[i]]) and torch.equal(full[i]["boxes"], boxes[i, : count[i]]) for i in range(n))
        # stale-shard protocol (yolort_amd/models/yolo.py PendingDetections.gathered): rank 1 marks its shard of the first exchange
        # stale (it has to re-run the batch locally); both ranks see it in their copy and both take the second round
        stale = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
        wrong = torch.full_like(scores[lo:hi], -7.0)   # what a truncated first pass might have left in the slab
        local = yd.pack_slab(boxes[lo:hi], wrong if rank == 1 else scores[lo:hi], labels[lo:hi], count[lo:hi], stale=stale)
        out = torch.empty(world * local.shape[0], local.shape[1])
        dist.all_gather_into_tensor(out, local)
        first = yd.unpack_slab(out, k)
        ok = ok and first[3].tolist() == [5, 0, 3, -1, -1, -1]
        calls = []
        def final():
            calls.append(1)
            return boxes[lo:hi], scores[lo:hi], labels[lo:hi], count[lo:hi]
        (b, s, l, c), second = yd.resolve_stale(first, final)
        ok = ok and second and len(calls) == 1 and torch.equal(b, boxes) and torch.equal(s, scores) and torch.equal(l, labels) and torch.equal(c, count)
        (b, s, l, c), second = yd.resolve_stale((b, s, l, c), final)   # nothing stale: no collective, no call
        ok = ok and not second and len(calls) == 1
        # unequal shards (N not divisible by G): padded to ceil(N / G) rows for the one fixed-shape collective, padding dropped afterwards
        n5 = 5
        lo5, hi5 = yd.shard_range(n5, rank, world)
        b, s, l, c = yd.all_gather_slab(boxes[lo5:hi5], scores[lo5:hi5], labels[lo5:hi5], count[lo5:hi5], global_n=n5)
        ok = ok and torch.equal(b, boxes[:n5]) and torch.equal(s, scores[:n5]) and torch.equal(l, labels[:n5]) and torch.equal(c, count[:n5])
        # every rank scores the GLOBAL batch from the gathered slab (the reference pickles per-rank results through two all_gathers,
        # data/coco_eval.py:225-226 -> data/distributed.py:6-49): identical numbers on both ranks, equal to a single-process evaluation
        from yolort_amd.utils.metrics import DetectionEvaluator
        gt = [{"boxes": boxes[i, : count[i]] + (0.5 if i % 2 else 0.0), "labels": labels[i, : count[i]]} for i in range(n)]
        ev = DetectionEvaluator(80)
        ev.update_from_slab(yd.all_gather_slab(boxes[lo:hi], scores[lo:hi], labels[lo:hi], count[lo:hi]), gt)
        single = DetectionEvaluator(80)
        single.update([{"boxes": boxes[i, : count[i]], "scores": scores[i, : count[i]], "labels": labels[i, : count[i]]} for i in range(n)], gt)
        res = ev.compute()
        ok = ok and res == single.compute() and res["AP"] > 0
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    from yolort_amd.dist import shard_range
    for n in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1
    assert shard_range(256, 3, 8) == (96, 128)  # BASELINE config 4: bs 256 over 8 GPUs, 32 per rank


def test_slab_pack_roundtrip_is_exact():
    from yolort_amd.dist import pack_slab, unpack_slab
    b, s = torch.rand(4, 300, 4), torch.rand(4, 300)
    l = torch.randint(0, 1 << 20, (4, 300))
    c = torch.tensor([300, 0, 17, 299], dtype=torch.int32)
    b2, s2, l2, c2 = unpack_slab(pack_slab(b, s, l, c), 300)
    assert torch.equal(b, b2) and torch.equal(s, s2) and torch.equal(l, l2) and torch.equal(c, c2)


@pytest.mark.timeout(120)
def test_all_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == [(0, True, (0, 3)), (1, True, (3, 6))]


# ---- round 5: the GLOBAL canvas of a sharded dynamic-shape stream (reference transform.py:307-314) and the stale protocol at world 4 ----------
def _canvas_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import yolov5_oracle as O
        from yolort_amd import dist as yd
        from yolort_amd.models import YOLOv5
        from workloads.synth import synth_images, synth_weights
        torch.set_num_threads(2)
        arch, S, thr, k = "yolov5_darknet_pan_n_r60", 96, 0.2, 300
        # four images whose shards have DIFFERENT local maxima: rank 0 sees two landscape images (local canvas 64 x 96 / 72 -> 96 x 96 after the resize rule below),
        # rank 1 a portrait and a small landscape one
        shapes = [(60, 96), (48, 96), (96, 56), (40, 72)]
        model = YOLOv5(arch=arch, size=(S, S), score_thresh=thr)
        sd = synth_weights(model.state_dict(), arch, seed=0, head_gain=0.6)
        imgs = [synth_images(1, h, w, seed=40 + i)[0] for i, (h, w) in enumerate(shapes)]
        lo, hi = yd.shard_range(len(imgs), rank, world)
        t = model.transform
        local = t.canvas_of(shapes[lo:hi])
        agreed = yd.agree_canvas(t, shapes[lo:hi])
        ok = agreed == t.canvas_of(shapes)            # == the canvas the reference pads the WHOLE list to
        with torch.no_grad():
            single = O.yolov5_forward(imgs, sd, size=(S, S), score_thresh=thr)                                   # the reference's semantics: one process, whole list
            mine_global = O.yolov5_forward(imgs[lo:hi], sd, size=(S, S), score_thresh=thr, fixed_shape=agreed)    # this rank's shard on the agreed canvas
            mine_local = O.yolov5_forward(imgs[lo:hi], sd, size=(S, S), score_thresh=thr)                         # ... on its own canvas (what round 4 did)
        full = yd.gather_detections(mine_global, k)
        same = all(torch.equal(f["labels"], s_["labels"]) and torch.equal(f["scores"], s_["scores"]) and torch.equal(f["boxes"], s_["boxes"]) for f, s_ in zip(full, single))
        ok = ok and len(full) == len(imgs) and same and sum(len(s_["scores"]) for s_ in single) >= 8
        full_local = yd.gather_detections(mine_local, k)
        differs = any(f["scores"].shape != s_["scores"].shape or not torch.equal(f["boxes"], s_["boxes"]) for f, s_ in zip(full_local, single))
        # the product's host geometry for this shard on the agreed canvas == the single-process geometry of the same images (sizes, pads, rescale rows)
        from yolort_amd.models.transform import rescale_params
        (hb, wb), sizes_all, pads_all = t.geometry(shapes)
        (hb2, wb2), sizes_sh, pads_sh = t.geometry(shapes[lo:hi], canvas=agreed)
        ok = ok and (hb2, wb2) == (hb, wb) and sizes_sh == sizes_all[lo:hi] and pads_sh == pads_all[lo:hi]
        ok = ok and [rescale_params((hb2, wb2), o) for o in shapes[lo:hi]] == [rescale_params((hb, wb), o) for o in shapes[lo:hi]]
        try:
            t.geometry(shapes, canvas=(32, 32))
            ok = False
        except ValueError:
            pass
        q.put((rank, bool(ok), bool(differs), tuple(local), tuple(agreed)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_dynamic_shape_stream_on_the_global_canvas_equals_the_single_process_result_world2_gloo():
    """Two ranks whose shards have different local canvases agree on the canvas of the whole list (one MAX all-reduce of two integers), letterbox onto it, and the
    gathered detections equal the single-process result on the whole list bit for bit; on their own canvases they do not (the reference pads to the maximum over the
    WHOLE list, transform.py:307-314).  The arithmetic here is the oracle's (CPU); that the HIP path honours `canvas=` is tests/test_e2e_gpu.py's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_canvas_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=280) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res
    assert res[0][3] != res[1][3] and res[0][4] == res[1][4] == (96, 96), res      # different local canvases, one agreed canvas
    assert res[0][2] or res[1][2], "per-rank canvases were expected to change some shard's detections"


def _stale4_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yolort_amd import dist as yd
        n, k = 8, 4
        lo, hi = yd.shard_range(n, rank, world)
        g = torch.Generator().manual_seed(99)
        boxes, scores, labels = torch.rand(n, k, 4, generator=g), torch.rand(n, k, generator=g), torch.randint(0, 80, (n, k), generator=g)
        count = torch.tensor([4, 0, 3, 1, 4, 2, 2, 4], dtype=torch.int32)
        stale_ranks = (1, 3)    # two of the four ranks have to re-run the batch locally
        stale = torch.tensor([1 if rank in stale_ranks else 0], dtype=torch.int32)
        wrong = torch.full_like(scores[lo:hi], -7.0)
        local = yd.pack_slab(boxes[lo:hi], wrong if rank in stale_ranks else scores[lo:hi], labels[lo:hi], count[lo:hi], stale=stale)
        out = torch.empty(world * local.shape[0], local.shape[1])
        dist.all_gather_into_tensor(out, local)
        first = yd.unpack_slab(out, k)
        ok = first[3].tolist() == [4, 0, -1, -1, 4, 2, -1, -1]
        calls = []

        def final():
            calls.append(1)
            return boxes[lo:hi], scores[lo:hi], labels[lo:hi], count[lo:hi]
        (b, s, l, c), second = yd.resolve_stale(first, final)
        ok = ok and second and len(calls) == 1 and torch.equal(b, boxes) and torch.equal(s, scores) and torch.equal(l, labels) and torch.equal(c, count)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_stale_shard_protocol_world4_two_stale_ranks_gloo():
    """the second round of the per-batch exchange (dist.resolve_stale) with four ranks of which two marked their shard stale: every rank sees both markers in its own
    copy of the first slab, all four enter the second all-gather exactly once, and the final slab is the global batch"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stale4_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == [(r, True) for r in range(4)]
