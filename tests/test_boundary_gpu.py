"""The reference's plug points on the MI355X: constructor injection of
`post_process=` / `head=` / `anchor_generator=` into YOLO (yolort/models/yolo.py:65-81,159-175), a pre-built `model=` handed
to YOLOv5 (yolov5.py:99,115-122), the operator-registry hook of INTEGRATION.md section 2, two fresh processes returning
bit-identical detections (pinned tile table), and more batches in flight than plan instances."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    from yolort_amd import _lib
    _lib.load(require_gpu=True)
    return torch.device("cuda:0")


def _np(d):
    return {k: (v.detach().float().cpu().numpy() if k != "labels" else v.detach().cpu().numpy()) for k, v in d.items()}


def _yolo(dev, thr=0.3, **hooks):
    """YOLO built the way the reference builds it (build_model, yolo.py:226-265) with optional injected modules"""
    from yolort_amd.models.backbone_utils import darknet_pan_backbone
    from yolort_amd.models.yolo import YOLO
    from workloads.synth import synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    backbone = darknet_pan_backbone("darknet_n_r6_0", 0.33, 0.25, version="r6.0")
    m = YOLO(backbone, 80, score_thresh=thr, nms_thresh=0.45, **hooks)
    sd = synth_weights(m.state_dict(), arch, seed=0, head_gain=1.0)
    m.load_state_dict(sd)
    return m.to(dev).half().eval(), sd


def test_injected_post_process_head_and_anchor_generator(dev):
    """`e.post is None` branch of YOLO._submit_entry: HIP backbone, then the injected modules are CALLED with the reference's
    tensor conventions -- head(List[NCHW]) -> List[(N,A,H,W,K)], anchor_generator(features) -> (grids, shifts),
    post_process(head_outputs, grids, shifts) -> List[Dict] -- and the result matches the oracle like the fused path does"""
    from oracle import yolov5_oracle as O
    from test_e2e_gpu import match_fraction
    from yolort_amd.models.anchor_utils import AnchorGenerator
    from yolort_amd.models.box_head import PostProcess, YOLOHead
    from yolort_amd.models.yolo import DEFAULT_ANCHORS
    from workloads.synth import synth_images
    calls = {"post": 0, "head": 0, "anchors": 0}

    class MyPost(PostProcess):
        def forward(self, head_outputs, grids, shifts):
            calls["post"] += 1
            assert len(head_outputs) == 3 and head_outputs[0].dim() == 5 and head_outputs[0].shape[1] == 3 and head_outputs[0].shape[-1] == 85
            assert grids[0].shape[-1] == 2 and shifts[0].shape[-1] == 2
            return super().forward(head_outputs, grids, shifts)

    class MyHead(YOLOHead):
        def forward(self, x):
            calls["head"] += 1
            return super().forward(x)

    class MyAnchors(AnchorGenerator):
        def forward(self, feature_maps):
            calls["anchors"] += 1
            return super().forward(feature_maps)

    x = synth_images(3, 256, 320, seed=12)
    ref_model, sd = _yolo(dev)
    fused = ref_model(x.to(dev).half())
    sdf = {"model." + k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        ref = O.yolo_forward(x, sdf, 0.3, 0.45, 300, p="model.")
    variants = {
        "post_process": dict(post_process=MyPost([8, 16, 32], 0.3, 0.45, 300)),
        "head": dict(head=MyHead([64, 128, 256], 3, [8, 16, 32], 80)),
        "anchor_generator": dict(anchor_generator=MyAnchors([8, 16, 32], DEFAULT_ANCHORS)),
    }
    for name, hooks in variants.items():
        m, _ = _yolo(dev, **hooks)
        assert not m.fused()
        out = m(x.to(dev).half())
        assert len(out) == 3 and list(out[0].keys()) == ["scores", "labels", "boxes"]
        for r, d, f in zip(ref, out, fused):
            assert len(r["scores"]) > 10
            frac, miou, _ = match_fraction(_np(r), _np(d), margin=0.03, thr=0.3)
            assert frac >= 0.9 and miou >= 0.9, (name, frac, miou)
            # hooks see fp16 head outputs (the reference's own GPU behaviour), the fused path fp32 logits: same detections up to that
            frac2, miou2, _ = match_fraction(_np(f), _np(d), margin=0.02, thr=0.3, score_tol=0.02)
            assert frac2 >= 0.95 and miou2 >= 0.97, (name, frac2, miou2)
    assert calls["post"] == 1 and calls["head"] == 1 and calls["anchors"] == 1, calls


def test_yolov5_accepts_a_prebuilt_model_and_rescales_after_a_post_process_hook(dev):
    """YOLOv5(model=...) (yolov5.py:99,115-122) + an injected post_process: boxes come back in ORIGINAL image coordinates via
    YOLOTransform.postprocess (yolov5.py:181), like the fused path's in-kernel rescale"""
    from test_e2e_gpu import match_fraction
    from yolort_amd.models import YOLOv5
    from yolort_amd.models.box_head import PostProcess
    from workloads.synth import synth_images

    class MyPost(PostProcess):
        pass

    hooked, _ = _yolo(dev, post_process=MyPost([8, 16, 32], 0.3, 0.45, 300))
    plain, _ = _yolo(dev)
    imgs = [synth_images(1, h, w, seed=70 + i)[0].to(dev).half() for i, (h, w) in enumerate([(200, 300), (333, 250)])]
    a = YOLOv5(model=hooked, size=(320, 320)).eval()(imgs)
    b = YOLOv5(model=plain, size=(320, 320)).eval()(imgs)
    for x, y in zip(a, b):
        assert len(y["scores"]) > 5
        frac, miou, _ = match_fraction(_np(y), _np(x), margin=0.02, thr=0.3, score_tol=0.02)
        assert frac >= 0.95 and miou >= 0.97, (frac, miou)


def test_post_process_hook_on_a_fixed_size_stream_runs_the_plan_from_the_planar_stem(dev):
    """identity-size compute-dtype batches feed the stem from the planar images (ops 0 / 0+1 outside the recorded plan) and never fill the
    NHWC4 canvas; the custom-hook branch must continue the plan from `first_op` instead of re-running op 0 on the empty canvas"""
    from test_e2e_gpu import match_fraction
    from yolort_amd.models import YOLOv5
    from yolort_amd.models.box_head import PostProcess

    class MyPost(PostProcess):
        pass

    from workloads.synth import synth_images
    hooked, _ = _yolo(dev, post_process=MyPost([8, 16, 32], 0.3, 0.45, 300))
    plain, _ = _yolo(dev)
    imgs = [synth_images(1, 320, 320, seed=90 + i)[0].to(dev).half() for i in range(2)]   # every image already is the canvas
    mh, mp = YOLOv5(model=hooked, size=(320, 320)).eval(), YOLOv5(model=plain, size=(320, 320)).eval()
    a, b = mh(imgs), mp(imgs)
    assert not hooked.fused() and plain.fused()
    for x, y in zip(a, b):
        assert len(y["scores"]) > 5
        frac, miou, _ = match_fraction(_np(y), _np(x), margin=0.02, thr=0.3, score_tol=0.02)
        assert frac >= 0.95 and miou >= 0.97, (frac, miou)


def test_operator_registry_hook(dev):
    """`yolort_amd::nms` registered through torch.library dispatches to ymi_batched_nms for CUDA
    tensors and returns the oracle's kept indices bit for bit"""
    from oracle import yolov5_oracle as O
    import yolort_amd.ops as ops
    lib = torch.library.Library("yolort_amd", "DEF")
    lib.define("nms(Tensor boxes, Tensor scores, Tensor labels, float iou) -> Tensor")
    lib.impl("nms", lambda b, s, l, iou: ops.batched_nms(b, s, l, iou), "CUDA")
    g = torch.Generator().manual_seed(3)
    n = 3000
    xy = torch.rand(n, 2, generator=g) * 600
    wh = torch.rand(n, 2, generator=g) * 120 + 4
    boxes = torch.cat([xy, xy + wh], 1)
    scores = (torch.rand(n, generator=g) * 64).floor() / 64          # heavy ties: order is decided by the stable index rule
    labels = torch.randint(0, 7, (n,), generator=g)
    keep = torch.ops.yolort_amd.nms(boxes.to(dev), scores.to(dev), labels.to(dev), 0.45)
    want = O.batched_nms(boxes, scores, labels, 0.45)
    assert keep.dtype == torch.int64 and torch.equal(keep.cpu(), want)
    with pytest.raises(Exception):
        torch.ops.yolort_amd.nms(boxes, scores, labels, 0.45)         # no CPU kernel registered: the op has no CPU fallback


_DET_SCRIPT = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
from yolort_amd.models import YOLOv5
from workloads.synth import synth_images, synth_weights
arch = "yolov5_darknet_pan_s_r60"
m = YOLOv5(arch=arch, score_thresh=0.25)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0))
m = m.to("cuda:0").half().eval()
dets = m.predict([im.to("cuda:0").half() for im in synth_images(4, 640, 640, seed=1)])
h = hashlib.sha256()
for d in dets:
    for k in ("scores", "labels", "boxes"):
        h.update(d[k].cpu().numpy().tobytes())
print("DIGEST", h.hexdigest(), sum(len(d["scores"]) for d in dets))
"""


_CPP_OP_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r)
from yolort_amd import torch_ext, ops
torch_ext.load()                                    # C++ TORCH_LIBRARY registration (yolort_amd/torch_ext/yolort_amd_ops.cpp): what a LibTorch program links
g = torch.Generator().manual_seed(5)
xy = torch.rand(3000, 2, generator=g) * 600
wh = torch.rand(3000, 2, generator=g) * 80 + 4
boxes = torch.cat([xy, xy + wh], 1).cuda()
scores = torch.rand(3000, generator=g).cuda()
labels = torch.randint(0, 7, (3000,), generator=g).cuda()
a = torch.ops.yolort_amd.batched_nms(boxes, scores, labels, 0.45)
b = ops.batched_nms(boxes, scores, labels, 0.45)     # the ctypes path of the Python package
assert a.dtype == torch.int64 and torch.equal(a, b), (a.shape, b.shape)
c = torch.ops.yolort_amd.nms(boxes, scores, 0.45)
d = ops.batched_nms(boxes, scores, torch.zeros_like(labels), 0.45)
assert torch.equal(c, d) and len(c) < len(a)
try:
    torch.ops.yolort_amd.nms(boxes.cpu(), scores.cpu(), 0.45)
    raise SystemExit("a CPU call must not succeed")
except (NotImplementedError, RuntimeError):
    pass
print("CPP_OP_OK", len(a), len(c))
"""


def test_cpp_operator_registration_for_libtorch_consumers(dev):
    """VERDICT r3 "missing" 6: the C++ `TORCH_LIBRARY(yolort_amd, ...)` registration a LibTorch program links instead of libtorchvision (reference
    test/tracing/CMakeLists.txt:5,13-18): built in-tree (g++ against the installed libtorch + libyolort_amd.so), loaded with torch.ops.load_library in a fresh
    process, `yolort_amd::batched_nms` / `::nms` equal to the package's own ctypes path index for index, no CPU kernel behind the op"""
    r = subprocess.run([sys.executable, "-c", _CPP_OP_SCRIPT % ROOT], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "CPP_OP_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])


def test_freeze_weights_serving_mode(dev):
    """YOLOv5.freeze_weights(): the plan key keeps the signature taken at the call (no per-batch walk over every tensor's version) -- identical detections, graph replay
    included; after freeze_weights(False) an in-place weight update is seen again (a new plan, other detections)"""
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    m = YOLOv5(arch=arch, size=(320, 320), score_thresh=0.3)
    m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=1.0))
    m = m.to(dev).half().eval()
    imgs = [im.to(dev).half() for im in synth_images(2, 320, 320, seed=61)]
    base = [_np(d) for d in m(imgs)]
    m.freeze_weights()
    m.model.use_graph = True
    for _ in range(3):
        got = [_np(d) for d in m(imgs)]
        for a, b in zip(base, got):
            assert all(np.array_equal(a[k], b[k]) for k in a)
    n_plans = sum(len(r) for r in m.model._ring.values())
    with torch.no_grad():
        m.model.head.head[0].bias.add_(0.5)          # frozen: the promise is the caller's, the cached plan (old packed weights) keeps serving
    got = [_np(d) for d in m(imgs)]
    assert all(np.array_equal(a[k], b[k]) for a, b in zip(base, got) for k in a) and sum(len(r) for r in m.model._ring.values()) == n_plans
    m.freeze_weights(False)
    m.model.use_graph = False
    got = [_np(d) for d in m(imgs)]                  # unfrozen: the update is part of the key again
    assert any(len(a["scores"]) != len(b["scores"]) or not np.array_equal(a["scores"], b["scores"]) for a, b in zip(base, got))


def test_post_process_as_a_graph_launch_equals_the_per_kernel_enqueues(dev):
    """round 6: ymi_plan_submit replays the post-process range (memsets, selection, sort, NMS, top-k) as a second captured graph on the side stream;
    `post_graph = False` enqueues the same launches one by one.  Detections identical over several batches (the graph is captured at the first and replayed), also with
    batches of different images alternating through the same plan instances"""
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_s_r60"
    m = YOLOv5(arch=arch, size=(320, 320), score_thresh=0.2)
    m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=0.8))
    m = m.to(dev).half().eval()
    batches = [[im.to(dev).half() for im in synth_images(4, 320, 320, seed=sd)] for sd in (11, 12, 13)]
    m.model.use_graph, m.model.post_graph = True, False
    base = [[_np(d) for d in m(b)] for b in batches]
    assert sum(len(d["scores"]) for b in base for d in b) > 50
    m.model.post_graph = True
    for _ in range(3):
        for b, want in zip(batches, base):
            got = [_np(d) for d in m(b)]
            assert all(np.array_equal(a[k], g[k]) for a, g in zip(want, got) for k in a)


def test_two_fresh_processes_return_bit_identical_detections(dev):
    """tiles come from the pinned table (yolort_amd/data/tiles_gfx950.json) or the library heuristic, never from timing at plan
    build (YOLORT_AMD_AUTOTUNE is opt-in), so the K accumulation order -- and every detection bit -- is the same in every process"""
    env = dict(os.environ)
    env.pop("YOLORT_AMD_AUTOTUNE", None)
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", _DET_SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1]
        outs.append(line)
    assert outs[0] == outs[1], outs
    assert int(outs[0].split()[-1]) > 100


def test_more_batches_in_flight_than_plan_instances(dev):
    """submitting more batches than `pipeline_depth` before collecting any must not let a later batch
    overwrite an uncollected one's results; result() is idempotent"""
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    m = YOLOv5(arch=arch, size=(160, 160), score_thresh=0.3)
    m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=1.0))
    m = m.to(dev).half().eval()
    m.model.pipeline_depth = 2
    batches = [[synth_images(1, 128, 160, seed=200 + 2 * i)[0].to(dev), synth_images(1, 160, 120, seed=201 + 2 * i)[0].to(dev)] for i in range(7)]
    sync = [m.forward(b) for b in batches]
    assert sum(len(d["scores"]) for r in sync for d in r) > 50 and len({len(r[0]["scores"]) for r in sync}) > 1   # batches differ
    pend = [m.forward_async(b) for b in batches]            # 7 in flight on 2 plan instances
    assert sum(len(r) for r in m.model._ring.values()) <= 2
    out = [p.result() for p in reversed(pend)][::-1]         # collected in reverse order on top
    for s_, o_, p in zip(sync, out, pend):
        assert p.result() is o_                              # idempotent
        for x, y in zip(s_, o_):
            assert torch.equal(x["labels"], y["labels"]) and torch.equal(x["scores"], y["scores"]) and torch.equal(x["boxes"], y["boxes"])


def test_capacity_redo_while_later_batches_are_in_flight(dev):
    """the default submit path (ymi_plan_begin / ymi_plan_submit, ONE completion event per plan instance) with `pipeline_depth` 2 and a candidate capacity
    that the very first batches overflow: each overflowing batch is redone (on grown buffers) while later batches already occupy the instances.  Every batch's detections
    equal those of a model that never overflowed, whatever the order of collection"""
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    def make(cap):
        m = YOLOv5(arch=arch, size=(160, 160), score_thresh=0.05)
        m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=1.0))
        m = m.to(dev).half().eval()
        m.model.pipeline_depth = 2
        m.model.cand_cap_per_image = cap
        return m
    batches = [[synth_images(1, 160, 160, seed=300 + 2 * i)[0].to(dev).half(), synth_images(1, 160, 160, seed=301 + 2 * i)[0].to(dev).half()] for i in range(6)]
    ref = make(1 << 16)
    want = [ref.forward(b) for b in batches]
    assert sum(len(d["scores"]) for r in want for d in r) > 100
    m = make(64)
    assert m.model.post_graph and m.model.use_graph                     # the default path is the one under test
    pend = [m.forward_async(b) for b in batches]                        # six submitted on two instances before any is collected; the first ones overflow 64 records
    got = [p.result() for p in pend[::2]] + [p.result() for p in pend[1::2]]
    got = [got[i // 2 + (3 if i % 2 else 0)] for i in range(6)]         # back into submission order
    assert m.model.cand_cap_per_image > 64                                # (the overflow was seen and the batch redone)
    for w_, g_ in zip(want, got):
        for x, y in zip(w_, g_):
            assert torch.equal(x["labels"], y["labels"]) and torch.equal(x["scores"], y["scores"]) and torch.equal(x["boxes"], y["boxes"])


def test_alternating_canvases_keep_their_plans(dev):
    """a variable-size stream alternates between canvases; each keeps its plan instances (small LRU) instead of
    rebuilding `pipeline_depth` plans per call"""
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    m = YOLOv5(arch=arch, size=(160, 160), score_thresh=0.3)
    m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=1.0))
    m = m.to(dev).half().eval()
    a = [synth_images(1, 128, 160, seed=1)[0].to(dev)]      # canvas 128x160
    b = [synth_images(1, 160, 96, seed=2)[0].to(dev)]       # canvas 160x96
    ra, rb = m.forward(a), m.forward(b)
    plans = {id(e.plan) for r in m.model._ring.values() for e in r}
    assert len(m.model._ring) == 2 and len(plans) == 2      # one lazily built instance per canvas
    for _ in range(3):
        xa, xb = m.forward(a), m.forward(b)
        assert torch.equal(xa[0]["boxes"], ra[0]["boxes"]) and torch.equal(xb[0]["boxes"], rb[0]["boxes"])
    assert {id(e.plan) for r in m.model._ring.values() for e in r} == plans   # nothing was rebuilt


def test_upstream_checkpoint_ingest_predicts_like_the_converted_state_dict(dev, tmp_path):
    """SURVEY 8f-1 on the GPU: a (synthetic) ultralytics-format checkpoint goes load_from_yolov5 -> predict; the detections
    equal, bit for bit, those of a model built the normal way from the converted state_dict (fp16-rounded like the reference,
    _checkpoint.py:81), and match the oracle run on that state_dict"""
    from oracle import yolov5_oracle as O
    from test_checkpoint_ingest import _write_fake_checkpoint
    from test_e2e_gpu import match_fraction
    from yolort_amd.models import YOLOv5, yolo as Y
    from yolort_amd.models._checkpoint import load_from_ultralytics
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    ref_sd = synth_weights(Y.__dict__[arch]().state_dict(), arch, seed=0, head_gain=1.0)
    path = str(tmp_path / "yolov5n_upstream_format.pt")
    _write_fake_checkpoint(path, ref_sd, p6=False)
    assert "models" not in sys.modules
    loaded = YOLOv5.load_from_yolov5(path, size=(320, 320), score_thresh=0.3).to(dev).half().eval()
    imgs_cpu = [synth_images(1, h, w, seed=80 + i)[0] for i, (h, w) in enumerate([(240, 320), (300, 210)])]
    imgs = [im.to(dev).half() for im in imgs_cpu]
    dets = loaded.predict(imgs)
    converted = load_from_ultralytics(path)["state_dict"]
    plain = YOLOv5(arch=arch, size=(320, 320), score_thresh=0.3)
    plain.model.load_state_dict(converted)
    plain = plain.to(dev).half().eval()
    dets2 = plain.predict(imgs)
    sdf = {"model." + k: v.float() for k, v in converted.items()}
    with torch.no_grad():
        ref = O.yolov5_forward(imgs_cpu, sdf, size=(320, 320), score_thresh=0.3)
    for a, b, r in zip(dets, dets2, ref):
        assert len(a["scores"]) > 10
        for k in ("scores", "labels", "boxes"):
            assert torch.equal(a[k], b[k])
        frac, miou, _ = match_fraction(_np(r), _np(a), margin=0.03, thr=0.3)
        assert frac >= 0.9 and miou >= 0.9, (frac, miou)


_GATHER_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%d), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from yolort_amd.models import YOLOv5
from workloads.synth import synth_images, synth_weights
arch = "yolov5_darknet_pan_n_r60"
m = YOLOv5(arch=arch, size=(160, 160), score_thresh=0.3)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=1.0))
m = m.to("cuda:0").half().eval()
batches = [[synth_images(1, 128, 160, seed=300 + i)[0].to("cuda:0"), synth_images(1, 160, 120, seed=400 + i)[0].to("cuda:0")] for i in range(4)]
for b in batches:
    m(b)   # candidate-capacity growth re-runs a batch locally, without a collective (PendingDetections.gathered raises for it): settle it first
m.model.enable_distributed_gather(force=True)
pend = [m.forward_async(b) for b in batches]
ok, total, per_batch = True, 0, []
for p in pend:
    dets = p.result()
    b, s, l, c = p.gathered()
    per_batch.append(int(c.max()))
    for i, d in enumerate(dets):
        k = int(c[i])
        checks = (k == len(d["scores"]), torch.equal(b[i, :k], d["boxes"]), torch.equal(s[i, :k], d["scores"]), torch.equal(l[i, :k], d["labels"]))
        if not all(checks):
            print("mismatch image", i, "count", k, len(d["scores"]), checks, l.dtype, d["labels"].dtype, s.dtype, d["scores"].dtype)
        ok = ok and all(checks)
    total += int(c.sum())
ok = ok and total > 0
print("total detections", total)
# a batch that has to be re-run locally AFTER the gather was switched on (candidate capacity too small): its shard of the first
# exchange is marked stale on the device, gathered() takes the second round (dist.resolve_stale) and returns the final results
crowded = max(range(len(batches)), key=lambda i: per_batch[i])   # > 64 detections in one image: more than 64 candidates for sure
ok = ok and per_batch[crowded] > 64
m.model.cand_cap_per_image = 64
p = m.forward_async(batches[crowded])
dets = p.result()
b, s, l, c = p.gathered()
print("second round taken:", p.second_round, "capacity now", m.model.cand_cap_per_image, "counts", c.tolist())
ok = ok and p.second_round and m.model.cand_cap_per_image > 64 and int(c.min()) >= 0
for i, d in enumerate(dets):
    k = int(c[i])
    ok = ok and k == len(d["scores"]) and k > 0 and torch.equal(b[i, :k], d["boxes"]) and torch.equal(s[i, :k], d["scores"]) and torch.equal(l[i, :k], d["labels"])
p = m.forward_async(batches[crowded])   # the grown capacity holds for the same batch: first round only
p.result(); p.gathered()
print("repeat: second round", p.second_round)
ok = ok and not p.second_round
print("GATHER_OK" if ok else "GATHER_MISMATCH")
dist.destroy_process_group()
"""


def test_slab_all_gather_is_enqueued_behind_the_post_process(dev):
    """SURVEY 8e / VERDICT r1 weak 12: with YOLO.enable_distributed_gather() the fixed-shape slab all-gather (RCCL, backend
    "nccl") is issued from the post-process stream of every batch; here on a world of one rank (the only GPU of the box), several
    batches in flight -- the gathered slab equals the local detections"""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    r = subprocess.run([sys.executable, "-c", _GATHER_SCRIPT % (ROOT, port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_the_topk_kernel_writes_the_wire_slab_itself(dev):
    """round 4: the packed wire slab of the multi-GPU gather -- (n, 6K + 1) fp32: boxes | scores | labels | count, slots past the count zeroed --
    is written by gather_topk_kernel next to the four output arrays (ymi_post_desc.out_slab); `dist.pack_slab` (a handful of torch launches) is off the serving
    path.  (1) normal batches: slab == pack_slab(outputs) on every live slot, zeros beyond, over several batches with different detection counts (a later batch with
    fewer detections must not leave the previous batch's entries behind); (2) a batch whose candidate capacity is too small: every row's count column reads
    SLAB_STALE (-1) on the device BEFORE the host has seen the status word -- the marker a collective enqueued behind the post-process carries to the other ranks;
    (3) status[4] carries the raw candidate count of the batch."""
    from yolort_amd import dist as ydist
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    m = YOLOv5(arch=arch, size=(320, 320), score_thresh=0.2)
    m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=1.0))
    m = m.to(dev).half().eval()
    k = 300
    for seed, thr in ((31, 0.2), (32, 0.35), (33, 0.2)):
        m.model.post_process.score_thresh = thr
        imgs = [im.to(dev).half() for im in synth_images(3, 320, 320, seed=seed)]
        for _ in range(3):   # (a first pass may overflow the candidate capacity: the host grows it and re-runs on another instance -- look at a pass that went through)
            pend = m.forward_async(imgs)
            dets = pend.result()
            torch.cuda.synchronize()
            post = pend.entry.post
            if int(post.status[1]) == 0:
                break
        assert int(post.status[1]) == 0
        want = ydist.pack_slab(post.boxes, post.scores, post.labels, post.count)
        slab = post.slab
        counts = post.count.tolist()
        assert counts == [len(d["scores"]) for d in dets] and max(counts) > 0
        for i, c in enumerate(counts):
            for lo, w in ((0, 4), (4 * k, 1), (5 * k, 1)):
                assert torch.equal(slab[i, lo: lo + w * c], want[i, lo: lo + w * c]), (seed, i, lo)
                assert float(slab[i, lo + w * c: lo + w * k].abs().sum()) == 0.0, (seed, i, lo)
            assert float(slab[i, 6 * k]) == c
        st = post.status.tolist()
        assert st[1] == 0 and st[4] >= st[0] >= sum(counts) and st[5] == 3, st
        b, s_, l, c_ = ydist.unpack_slab(slab, k)
        assert torch.equal(c_.cpu(), post.count.cpu()) and torch.equal(l[0, : counts[0]], post.labels[0, : counts[0]])
    # (2) the stale marker: a candidate capacity of 64 per image overflows; look at the first-pass slab of the instance BEFORE collecting the batch
    m2 = YOLOv5(arch=arch, size=(320, 320), score_thresh=0.05)
    m2.load_state_dict(synth_weights(m2.state_dict(), arch, seed=0, head_gain=1.0))
    m2 = m2.to(dev).half().eval()
    m2.model.cand_cap_per_image = 64
    imgs = [im.to(dev).half() for im in synth_images(3, 320, 320, seed=34)]
    pend = m2.forward_async(imgs)
    pend.event.synchronize()
    post = pend.entry.post
    assert int(post.status[1]) != 0
    assert torch.equal(post.slab[:, 6 * k].cpu(), torch.full((3,), float(ydist.SLAB_STALE)))
    dets = pend.result()   # the host grows the capacity and re-runs: final results, final slab
    assert sum(len(d["scores"]) for d in dets) > 0 and m2.model.cand_cap_per_image > 64


def test_bench_two_ranks_control_flow_on_one_gpu(dev):
    """bench.py's N > 1 path (per-rank shard, warm-up before the gather is switched on, slab all-gather per batch, gathered() for
    every batch, barrier + max-over-ranks timing, one JSON line from rank 0) run as TWO ranks under torch.distributed.run.  The box
    has one GPU, so both ranks use cuda:0 and the process group is gloo (RCCL refuses two ranks on one device) -- the collectives
    are the same calls, only the transport differs; the RCCL transport itself is covered by the one-rank test above."""
    import json
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, YOLORT_AMD_BENCH_BACKEND="gloo", YOLORT_AMD_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c1", "--steps", "6", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-800:], r.stderr[-2500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"].startswith("dp2") and d["config"]["gather_second_rounds_rank0"] == 0
    assert "cpu_baseline" not in d   # rank 0 at N = 1 only


def test_exported_plan_replays_without_python_objects(dev, tmp_path):
    """round 6: `YOLO.export_plan` writes the recorded plan -- descriptors with (region, offset) pointers + the constant regions'
    contents -- and `ymi_plan_import` rebuilds it in fresh device memory with nothing but the C ABI: the replay on the same letterboxed canvas returns the same detection
    arrays bit for bit.  (A C++ consumer does exactly what this test does through ctypes: INTEGRATION.md section 5.)"""
    import ctypes as C
    from yolort_amd import _lib
    from yolort_amd._lib import PlanRegion, TAG_BOXES, TAG_INPUT, TAG_LABELS, TAG_RESCALE, TAG_SCORES, TAG_STATUS_COUNT
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    lib = _lib.load(require_gpu=True)
    for arch, S in (("yolov5_darknet_pan_n_r60", 160), ("yolov5_darknet_pan_s_r60", 320)):   # (s at 320: its C3 blocks of 64 / 128 hidden channels run through the strip kernel's weight streams)
        m = YOLOv5(arch=arch, size=(S, S), score_thresh=0.1)
        m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=0.8))
        m = m.to(dev).half().eval()
        imgs = [synth_images(1, S - 40, S, seed=21)[0].to(dev).half(), synth_images(1, S, S - 64, seed=22)[0].to(dev).half()]   # ragged sizes: the letterbox path (canvas input)
        dets = m.predict(imgs)
        torch.cuda.synchronize()
        e = next(iter(m.model._entries.values()))
        assert sum(len(d["scores"]) for d in dets) > 0
        path = str(tmp_path / f"{arch}.ymiplan")
        info = m.model.export_plan(path, e.x.n, e.x.h, e.x.w, dev)
        assert os.path.getsize(path) > 1000 and info["ops"] == e.plan.num_ops
        # ---- the consumer: C ABI only ----
        plan2 = C.c_void_p()
        regs = (PlanRegion * 4096)()
        nreg = C.c_int(0)
        assert lib.ymi_plan_import(path.encode(), C.byref(plan2), regs, 4096, C.byref(nreg)) == 0, lib.ymi_last_error()
        assert nreg.value == info["regions"]
        by_tag = {regs[i].tag: regs[i] for i in range(nreg.value) if regs[i].tag}
        assert set(by_tag) >= {TAG_INPUT, TAG_RESCALE, TAG_BOXES, TAG_SCORES, TAG_LABELS, TAG_STATUS_COUNT}

        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes, hip.hipMemcpy.restype = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int], C.c_int

        def dev_copy(dst_ptr, src):   # device -> device, as the consumer would
            torch.cuda.synchronize()
            assert hip.hipMemcpy(dst_ptr, src.data_ptr(), src.numel() * src.element_size(), 3) == 0   # hipMemcpyDeviceToDevice

        dev_copy(by_tag[TAG_INPUT].base, e.x.base)        # the letterboxed canvas of the batch above
        dev_copy(by_tag[TAG_RESCALE].base, e.rescale)
        s = torch.cuda.Stream()
        assert lib.ymi_plan_run(plan2, 0, -1, 0, C.c_void_p(s.cuda_stream)) == 0, lib.ymi_last_error()
        s.synchronize()

        def read(tag, like):
            out = torch.empty_like(like)
            assert hip.hipMemcpy(out.data_ptr(), by_tag[tag].base, out.numel() * out.element_size(), 3) == 0
            return out

        boxes, scores, labels, sc = read(TAG_BOXES, e.post.boxes), read(TAG_SCORES, e.post.scores), read(TAG_LABELS, e.post.labels), read(TAG_STATUS_COUNT, e.post.status_count)
        torch.cuda.synchronize()
        cnt = sc[8:].tolist()
        assert cnt == [len(d["scores"]) for d in dets], (cnt, [len(d["scores"]) for d in dets])
        for i, d in enumerate(dets):
            k = cnt[i]
            assert torch.equal(boxes[i, :k], d["boxes"]) and torch.equal(scores[i, :k], d["scores"]) and torch.equal(labels[i, :k], d["labels"])
        lib.ymi_plan_destroy(plan2)
