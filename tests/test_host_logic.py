"""Host-side logic of the product (no GPU): letterbox geometry against the reference goldens, API
surface/contracts of the mirrored modules, and the 'no CPU fallback' behaviour."""
import json
import os

import numpy as np
import pytest
import torch


def test_letterbox_geometry_matches_reference(golden_dir):
    from yolort_amd.models.transform import YOLOTransform, pad_offset, resized_hw
    g = json.load(open(os.path.join(golden_dir, "letterbox.json")))
    n = 0
    for c in g["cases"]:
        if c.get("mixed"):
            t = YOLOTransform(640, 640, fixed_shape=tuple(c["fixed"]) if c["fixed"] else None)
            (hb, wb), sizes, pads = t.geometry([(1080, 810), (480, 640), (720, 1280)])
            assert [3, 3, hb, wb] == c["canvas"] and [list(s) for s in sizes] == c["image_sizes"]
            continue
        t = YOLOTransform(c["S"], c["S"], size_divisible=c["stride"])
        (hb, wb), sizes, pads = t.geometry([tuple(c["hw"])])
        assert list(sizes[0]) == c["resized"], c
        assert [hb, wb] == c["canvas"], c
        n += 1
    assert n == 40
    # SURVEY.md Appendix B rounding traps, spelled out
    assert resized_hw(1281, 1279, 640.0, 640.0) == (639, 639)
    assert resized_hw(375, 500, 640.0, 640.0) == (480, 640)
    assert resized_hw(100, 37, 1280.0, 1280.0) == (1279, 473)
    assert pad_offset(448, 427) == 10 and pad_offset(640, 427) == 106
    t = YOLOTransform(640, 640, fixed_shape=(640, 640))
    (_, _), sizes, pads = t.geometry([(427, 640)])
    assert pads[0][0] == g["fixed_427"]["first_row"] and pads[0][0] + sizes[0][0] - 1 == g["fixed_427"]["last_row"]
    assert float(np.float32(t.fill_color)) == g["fill"]


def test_rescale_matches_reference_scale_coords(golden_dir):
    from yolort_amd.models.transform import rescale_params, scale_coords
    g = json.load(open(os.path.join(golden_dir, "letterbox.json")))
    b = scale_coords(torch.tensor([[100.0, 100.0, 200.0, 200.0]]), (640, 480), (1080, 810))
    assert [float(v) for v in b.flatten()] == g["scale_coords"]
    b = scale_coords(torch.tensor([[10.5, 20.25, 300.75, 333.0]]), (384, 640), (720, 1280))
    assert [float(v) for v in b.flatten()] == g["scale_coords2"]
    gain, px, py = rescale_params((640, 480), (1080, 810))
    assert abs(gain - 0.59259) < 1e-5 and px == 0.0 and py == 0.0


def test_make_divisible_reference_values():
    """values of reference test/test_models_utils.py:16-37"""
    from yolort_amd.models._utils import _make_divisible
    assert [_make_divisible(v, 8) for v in (16, 32 * 0.25, 64 * 0.5, 768 * 0.75, 1024 * 1.25, 30)] == [16, 8, 32, 576, 1280, 32]
    assert _make_divisible(10, 8) == 16 and _make_divisible(3, 8) == 8


def test_anchor_generator_known_answers(golden_dir):
    """reference test/test_models_anchor_utils.py:14-30"""
    from yolort_amd.models.anchor_utils import AnchorGenerator
    z = np.load(os.path.join(golden_dir, "anchors_decode.npz"))
    grids, shifts = AnchorGenerator([4], [[6, 14]])([torch.rand(1, 3, 2, 2)])
    np.testing.assert_array_equal(grids[0].numpy(), z["grids"])
    np.testing.assert_array_equal(shifts[0].numpy(), z["shifts"])


def test_api_surface_and_kwargs():
    import yolort_amd.models as M
    for name in ("YOLO", "YOLOv5", "yolov5n", "yolov5n6", "yolov5s", "yolov5s6", "yolov5m", "yolov5m6", "yolov5l", "yolov5ts"):
        assert hasattr(M, name)
    m = M.yolov5s(score_thresh=0.35, nms_thresh=0.5, detections_per_img=100, size=(320, 416), fill_color=0)
    pp = m.model.post_process
    assert (pp.score_thresh, pp.nms_thresh, pp.detections_per_img) == (0.35, 0.5, 100)
    assert (m.transform.min_size, m.transform.max_size, m.transform.fill_color) == (320, 416, 0.0)
    assert len(m.state_dict()) == 348 and all(k.startswith("model.") for k in m.state_dict())
    assert M.yolov5n6().transform.size_divisible == 64
    assert M.yolov5n6().model.anchor_generator.strides == [8, 16, 32, 64]
    with pytest.raises(NotImplementedError):
        M.yolov5n(upstream_version="r4.0")
    with pytest.raises(NotImplementedError):
        M.yolov5ts()
    with pytest.raises(ValueError):
        M.YOLO(torch.nn.Identity(), 80)  # backbone without out_channels (reference yolo.py:83-88)
    # defaults of the reference (yolo.py:77-79)
    d = M.YOLOv5(arch="yolov5_darknet_pan_n_r60").model.post_process
    assert (d.score_thresh, d.nms_thresh, d.detections_per_img) == (0.005, 0.45, 300)


@pytest.mark.parametrize("version", ["r4.0", "r3.1"])
def test_legacy_release_models_mirror_the_reference(version):
    """round 5: the r3.1 / r4.0 releases -- Focus stem, BottleneckCSP (r3.1) / C3 (r4.0) blocks, the neck that opens with a block.  Same state_dict keys and
    shapes as the reference's models (live comparison when /root/reference is importable; the counts are pinned either way), the reference's activation modules, and the
    Focus stem's 6 x 6 stride-2 form equal to Conv(12, c, 3) over `focus_transform`."""
    import torch.nn.functional as F
    import yolort_amd.models as M
    from yolort_amd.models import yolo
    from yolort_amd.v5 import Focus, focus_transform, space_to_depth
    tag = version.replace(".", "").replace("r", "r")
    counts = {"r4.0": {"s": 360, "m": 504, "l": 648}, "r3.1": {"s": 368, "m": 512, "l": 656}}[version]
    for size in ("s", "m", "l"):
        mine = yolo.__dict__[f"yolov5_darknet_pan_{size}_{tag}"]().state_dict()
        assert len(mine) == counts[size], (size, len(mine))
        try:
            from oracle.reference_loader import load_reference, reference_available
        except ImportError:
            continue
        if reference_available():
            ref = load_reference().models.yolo.__dict__[f"yolov5_darknet_pan_{size}_{tag}"]().state_dict()
            assert list(ref) == list(mine), size                                               # same keys in the same order
            assert all(tuple(ref[k].shape) == tuple(mine[k].shape) and ref[k].dtype == mine[k].dtype for k in ref), size
    m = M.yolov5s(upstream_version=version, score_thresh=0.3)
    body = m.model.backbone.body
    assert isinstance(body["0"], Focus) and type(body["2"]).__name__ == ("C3" if version == "r4.0" else "BottleneckCSP")
    assert type(body["1"].act).__name__ == ("SiLU" if version == "r4.0" else "Hardswish")
    assert type(m.model.backbone.pan.inner_blocks[0]).__name__ == type(body["2"]).__name__        # reference path_aggregation_network.py:111-112
    # Focus == Conv(3, c, 6, 2, 2) with the rearranged weights (v5/models/common.py Focus.stem_weight)
    torch.manual_seed(0)
    f = Focus(3, 8, k=3, version=version).double()
    x = torch.rand(2, 3, 12, 20, dtype=torch.float64)
    assert torch.equal(focus_transform(x), space_to_depth(x))                                     # reference test/test_models_common.py:6-11
    a = F.conv2d(focus_transform(x), f.conv.conv.weight, None, 1, 1)
    b = F.conv2d(x, f.stem_weight(), None, 2, 2)
    assert a.shape == b.shape and (a - b).abs().max().item() < 1e-12
    with pytest.raises(NotImplementedError):
        yolo.yolov5_darknet_tan_s_r40()


@pytest.mark.parametrize("n, b, h, w", [(1, 3, 480, 640), (4, 3, 416, 320), (4, 3, 320, 416)])
def test_space_to_depth(n, b, h, w):
    """reference test/test_models_common.py:6-11, unchanged but for the import"""
    from yolort_amd.v5 import focus_transform, space_to_depth
    tensor_input = torch.rand((n, b, h, w))
    out1 = focus_transform(tensor_input)
    out2 = space_to_depth(tensor_input)
    torch.testing.assert_close(out2, out1)


def test_c_pass_over_the_image_list_agrees_with_the_per_image_reads():
    """torch_ext/sig_ext.cpp `images()`: the checks YOLOv5.forward_async / Plan.stem_planar_ok / Plan.stem_from_planar make per image (3-d, device, dtype, shape,
    contiguity, 16-byte alignment, data_ptr), in one pass -- against the same facts read through the tensors' Python attributes (CPU tensors here: device index -1)"""
    import struct

    from yolort_amd import hipmodule
    ext = hipmodule._SIG_EXT
    if ext is None:
        pytest.skip("yolort_amd/lib/_ymi_sig.so is not built (python -m yolort_amd.torch_ext)")
    wide = torch.rand(3, 16, 32)
    cases = {
        "uniform": [torch.rand(3, 16, 24) for _ in range(5)],
        "mixed shapes": [torch.rand(3, 16, 24), torch.rand(3, 8, 24), torch.rand(3, 16, 24)],
        "mixed dtypes": [torch.rand(3, 8, 8), torch.rand(3, 8, 8).half()],
        "strided view": [wide[:, :, ::2], wide[:, :, 1::2]],
        "interleaved uint8": [torch.zeros(16, 24, 3, dtype=torch.uint8)] * 2,
        "a 2-d tensor": [torch.rand(3, 8, 8), torch.rand(8, 8)],
        "misaligned": [torch.rand(3 * 8 * 8 + 1)[1:].view(3, 8, 8)],
    }
    for name, ims in cases.items():
        all3, dev, same, uniform, shapes, contig, aligned, ptrs = ext.images(ims)
        assert all3 == all(t.dim() == 3 for t in ims), name
        assert dev == -1, name                                                        # nothing here lives on a GPU
        assert struct.unpack(f"{len(ims)}Q", ptrs) == tuple(t.data_ptr() for t in ims), name
        if all3:
            assert same == (len({t.dtype for t in ims}) == 1), name
            assert contig == all(t.is_contiguous() for t in ims), name
            assert aligned == all(t.data_ptr() % 16 == 0 for t in ims), name
            if len({tuple(t.shape) for t in ims}) == 1:
                assert uniform == tuple(ims[0].shape) and shapes is None, name
            else:
                assert uniform is None and shapes == tuple(tuple(t.shape) for t in ims), name
    assert not cases["misaligned"][0].data_ptr() % 16 == 0 and cases["strided view"][0].is_contiguous() is False
    with pytest.raises(TypeError):
        ext.images([torch.rand(3, 4, 4), "not a tensor"])
    with pytest.raises(TypeError):
        ext.images((torch.rand(3, 4, 4),))   # a list is required (forward_async builds one)
    assert ext.images([])[3] is None


def test_no_cpu_fallback_anywhere():
    from yolort_amd._lib import YmiError
    from yolort_amd.models import yolov5n
    from yolort_amd.v5 import C3, Conv
    m = yolov5n().eval()
    with pytest.raises(YmiError):
        m.predict(torch.rand(3, 64, 64))
    with pytest.raises(YmiError):
        m.model(torch.rand(1, 3, 64, 64))
    with pytest.raises(YmiError):
        Conv(8, 16, 3).eval()(torch.rand(1, 8, 16, 16))
    with pytest.raises(YmiError):
        C3(16, 16).eval()(torch.rand(1, 16, 8, 8))
    with pytest.raises(NotImplementedError):
        yolov5n().train()(torch.rand(1, 3, 64, 64))


def test_collate_images_contract():
    """reference yolov5.py:230-262 (device/dtype follow the model; unsupported types raise)"""
    from yolort_amd.models import yolov5n
    m = yolov5n().half()
    out = m.collate_images([torch.rand(3, 8, 8), torch.rand(3, 4, 4)], m.default_loader)
    assert all(t.dtype == torch.float16 for t in out) and len(out) == 2
    assert m.collate_images(torch.zeros(3, 8, 8, dtype=torch.uint8), m.default_loader)[0].dtype == torch.uint8
    with pytest.raises(NotImplementedError):
        m.collate_images(3.14, m.default_loader)


def test_synth_weights_are_deterministic():
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_weights
    arch = "yolov5_darknet_pan_n_r60"
    t = YOLOv5(arch=arch).state_dict()
    a, b = synth_weights(t, arch, seed=0), synth_weights(t, arch, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert float(a["model.backbone.body.0.bn.running_var"].mean()) != 1.0  # calibrated statistics were loaded


def test_bench_coco_ap_metric():
    """bench.py's 'mAP vs ref' (SURVEY.md 8d): identical detections score 1.0, shifted boxes score less, missing classes count"""
    import importlib.util
    import os

    import numpy as np

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(0)

    def mk(n):
        xy = rng.random((n, 2)) * 500
        wh = rng.random((n, 2)) * 80 + 20
        return {"boxes": np.concatenate([xy, xy + wh], 1), "scores": rng.random(n), "labels": rng.integers(0, 5, n)}

    refs = [mk(50), mk(40)]
    assert bench.coco_ap(refs, refs, 5) == 1.0
    shifted = [{**r, "boxes": r["boxes"] + 3.0} for r in refs]
    ap = bench.coco_ap(refs, shifted, 5)
    assert 0.3 < ap < 0.9
    dropped = [{k: v[r["labels"] != 0] for k, v in r.items()} for r in refs]   # class 0 never detected
    assert bench.coco_ap(refs, dropped, 5) < 0.85
    assert bench.coco_ap([{"boxes": np.zeros((0, 4)), "scores": np.zeros(0), "labels": np.zeros(0, int)}], [mk(3)], 5) is None


def test_detection_evaluator_surface():
    """update()/compute() accumulator over in-memory ground truth (the reference's COCOEvaluator surface):
    perfect detections score 100, shards merge, half the boxes missing halves the recall-limited AP"""
    import numpy as np
    import torch

    from yolort_amd.utils.metrics import DetectionEvaluator

    rng = np.random.default_rng(3)

    def mk(n):
        xy = rng.random((n, 2)) * 500
        wh = rng.random((n, 2)) * 80 + 20
        return {"boxes": torch.tensor(np.concatenate([xy, xy + wh], 1)), "scores": torch.tensor(rng.random(n)), "labels": torch.tensor(rng.integers(0, 4, n))}

    imgs = [mk(40), mk(30), mk(20)]
    tg = [{"boxes": d["boxes"], "labels": d["labels"]} for d in imgs]
    a, b = DetectionEvaluator(4), DetectionEvaluator(4)
    a.update(imgs[:2], tg[:2])
    b.update(imgs[2:], tg[2:])
    a.merge(b)
    r = a.compute()
    assert r["AP"] == 100.0 and r["AP50"] == 100.0 and r["AP75"] == 100.0
    half = DetectionEvaluator(4)
    half.update([{k: v[::2] for k, v in d.items()} for d in imgs], tg)
    h = half.compute()
    assert 35.0 < h["AP50"] < 65.0
    assert DetectionEvaluator(4).compute()["AP"] == -1.0


def test_coco_ap_known_answer_from_the_published_protocol():
    """hand-computed vector for the COCO protocol (pycocotools cocoeval.accumulate: greedy score-ordered matching, precision
    envelope, 101 recall points with searchsorted 'left', mean over IoU .50:.05:.95): one class, 2 ground-truth boxes,
    detections d1 (0.9, IoU 1.0 with g1), d2 (0.8, no overlap), d3 (0.7, IoU 0.8 with g2).
      thresholds .50 ... .80 (7 of 10): TP FP TP -> recall .5 .5 1, precision 1 .5 2/3 -> envelope 1 2/3 2/3
                                        AP = (51 * 1 + 50 * 2/3) / 101
      thresholds .85 .90 .95:           TP FP FP -> recall stays .5 -> AP = 51 / 101
    """
    import numpy as np

    from yolort_amd.utils.metrics import COCO_IOU_THRS, DetectionEvaluator, coco_ap
    assert len(COCO_IOU_THRS) == 10 and COCO_IOU_THRS[4] == 0.7 and abs(COCO_IOU_THRS[-1] - 0.95) < 1e-12
    g = {"boxes": np.array([[0, 0, 10, 10], [20, 20, 30, 30]], np.float32), "labels": np.array([0, 0]), "scores": np.ones(2, np.float32)}
    d = {"boxes": np.array([[0, 0, 10, 10], [50, 50, 60, 60], [20, 20, 30, 28]], np.float32), "labels": np.array([0, 0, 0]),
         "scores": np.array([0.9, 0.8, 0.7], np.float32)}
    want = (7 * (51 + 50 * 2 / 3) / 101 + 3 * 51 / 101) / 10
    assert abs(coco_ap([g], [d], num_classes=1) - want) < 1e-9
    # maxDets: only the best `max_dets` detections of an image count (COCO: 100) -- with max_dets=2, d3 is never seen
    assert abs(coco_ap([g], [d], num_classes=1, max_dets=2) - 51 / 101) < 1e-9
    ev = DetectionEvaluator(1)
    assert ev.max_dets == 100
    ev.update([d], [{"boxes": g["boxes"], "labels": g["labels"]}])
    out = ev.compute()
    assert abs(out["AP"] - 100 * want) < 1e-6 and abs(out["AP50"] - 100 * (51 + 50 * 2 / 3) / 101) < 1e-6 and abs(out["AP75"] - 100 * (51 + 50 * 2 / 3) / 101) < 1e-6


def test_coco_max_dets_is_per_image_and_category():
    """pycocotools cuts `_dts[imgId, catId]` at maxDets -- per (image, category).  150 high-scoring class-0 false positives
    must not push a correct, low-scoring class-1 detection out of the evaluation (a per-image cap scored it 0.0)."""
    import numpy as np

    from yolort_amd.utils.metrics import coco_ap
    fp = np.tile(np.array([[500, 500, 520, 520]], np.float32), (150, 1)) + np.arange(150, dtype=np.float32)[:, None]
    g = {"boxes": np.array([[0, 0, 10, 10]], np.float32), "labels": np.array([1]), "scores": np.ones(1, np.float32)}
    d = {"boxes": np.concatenate([fp, g["boxes"]]), "labels": np.array([0] * 150 + [1]), "scores": np.concatenate([np.linspace(0.99, 0.5, 150), [0.1]]).astype(np.float32)}
    assert coco_ap([g], [d], num_classes=2, max_dets=100) == 1.0
    assert coco_ap([g], [d], num_classes=2) == 1.0
    # ... and within a category the cut is by score: 101 class-1 detections, the correct one scoring lowest, max_dets 100 -> never seen
    d2 = {"boxes": np.concatenate([fp[:100], g["boxes"]]), "labels": np.ones(101, int), "scores": np.concatenate([np.linspace(0.99, 0.5, 100), [0.1]]).astype(np.float32)}
    assert coco_ap([g], [d2], num_classes=2, max_dets=100) == 0.0
    assert coco_ap([g], [d2], num_classes=2) > 0.0


def test_coco_area_ranges_known_answer():
    """COCO's small / medium / large lines (pycocotools evaluateImg: ground truth outside the range is ignored, a detection matched to it is
    ignored, an unmatched detection outside the range is ignored).  One class: g_small 20x20 (area 400), g_large 100x100 (10 000);
    d1 0.9 == g_large, d2 0.8 == g_small, d3 0.7 a 200x200 false positive, d4 0.6 a 10x10 false positive.
      all:    TP TP FP FP -> AP 1.0                      small:  d1 ignored (matched to an ignored box), d2 TP, d3 ignored (unmatched, area out of
      range), d4 FP after full recall -> AP 1.0          large:  d1 TP, d2 ignored, d3 FP after full recall, d4 ignored -> 1.0
      medium: no ground truth in range -> not scored (-1)
    and with d2 moved off its box:  small: d2' (unmatched, area 400 in range) FP, recall 0 -> AP 0;  large unchanged"""
    import numpy as np

    from yolort_amd.utils.metrics import COCO_AREA_RNG, DetectionEvaluator, coco_ap
    g = {"boxes": np.array([[0, 0, 20, 20], [100, 100, 200, 200]], np.float32), "labels": np.zeros(2, int), "scores": np.ones(2, np.float32)}
    d = {"boxes": np.array([[100, 100, 200, 200], [0, 0, 20, 20], [300, 300, 500, 500], [600, 600, 610, 610]], np.float32), "labels": np.zeros(4, int),
         "scores": np.array([0.9, 0.8, 0.7, 0.6], np.float32)}
    assert coco_ap([g], [d], 1) == 1.0
    assert coco_ap([g], [d], 1, area_rng=COCO_AREA_RNG["small"]) == 1.0
    assert coco_ap([g], [d], 1, area_rng=COCO_AREA_RNG["large"]) == 1.0
    assert coco_ap([g], [d], 1, area_rng=COCO_AREA_RNG["medium"]) is None
    d2 = dict(d, boxes=d["boxes"].copy())
    d2["boxes"][1] = [700, 700, 720, 720]
    assert coco_ap([g], [d2], 1, area_rng=COCO_AREA_RNG["small"]) == 0.0
    assert coco_ap([g], [d2], 1, area_rng=COCO_AREA_RNG["large"]) == 1.0
    # small range where the false positive d4 (area 100, in range) outranks the true positive: precision 1/2 at full recall
    d3 = dict(d, scores=np.array([0.9, 0.5, 0.7, 0.6], np.float32))
    assert abs(coco_ap([g], [d3], 1, area_rng=COCO_AREA_RNG["small"]) - 0.5) < 1e-9
    ev = DetectionEvaluator(1)
    ev.update([d], [{"boxes": g["boxes"], "labels": g["labels"]}])
    out = ev.compute()
    assert out["APs"] == 100.0 and out["APl"] == 100.0 and out["APm"] == -1.0 and out["AP"] == 100.0


def test_evaluator_consumes_the_gathered_slab():
    import numpy as np
    import torch

    from yolort_amd import dist as ydist
    from yolort_amd.utils.metrics import DetectionEvaluator
    rng = np.random.default_rng(0)

    def mk(n):
        xy = rng.random((n, 2)) * 300
        wh = rng.random((n, 2)) * 60 + 10
        return {"boxes": torch.tensor(np.concatenate([xy, xy + wh], 1), dtype=torch.float32), "scores": torch.tensor(rng.random(n), dtype=torch.float32),
                "labels": torch.tensor(rng.integers(0, 3, n))}

    dets = [mk(7), mk(0), mk(12)]
    slab = ydist.dets_to_slab(dets, k=16)
    a, b = DetectionEvaluator(3), DetectionEvaluator(3)
    tg = [{"boxes": d["boxes"], "labels": d["labels"]} for d in dets]
    a.update(dets, tg)
    b.update_from_slab(slab, tg)
    assert a.compute() == b.compute() and b.compute()["AP"] == 100.0
    stale = (slab[0], slab[1], slab[2], torch.tensor([7, ydist.SLAB_STALE, 12], dtype=torch.int32))
    import pytest
    with pytest.raises(ValueError):
        DetectionEvaluator(3).update_from_slab(stale, tg)


def test_pinned_resident_weight_tiles_are_validated_against_the_launch():
    """engine.Plan._pinned_ok: a table entry >= 132 is keyed by shape only; the kernels behind tiles 132-138 also require SiLU, no chained conv, 32-bit
    offsets ... -- a launch that does not meet them must fall through to the general tiles instead of failing at plan build, and an explicit opt-out wins over the table"""
    from yolort_amd import engine
    from yolort_amd._lib import ACT_NONE, ACT_SILU, ConvDesc, YMI_F16

    def desc(cin, cout, s, act, h=80):
        d = ConvDesc()
        d.n, d.h, d.w_in, d.cin, d.x_cstride = 32, h, h, cin, cin
        d.ho = d.wo = (h + 2 - 3) // s + 1
        d.cout, d.cout_pad, d.y_cstride, d.res_cstride = cout, (cout + 31) // 32 * 32, cout, 0
        d.kh = d.kw = 3
        d.sh = d.sw = s
        d.ph = d.pw = 1
        d.k_pad, d.act, d.dtype, d.out_dtype = 9 * cin, act, YMI_F16, YMI_F16
        d.zeros = 1   # (a non-null zero page)
        return d

    p = engine.Plan.__new__(engine.Plan)
    p.res3x3, p.rw2, p.rw3, p.rs = 2, 1, False, False
    assert p._pinned_ok(desc(64, 64, 1, ACT_SILU), 133) and p._pinned_ok(desc(64, 64, 1, ACT_SILU), 132)
    assert not p._pinned_ok(desc(64, 64, 1, ACT_NONE), 133)             # Conv(act=False) on a pinned shape: tile 133's launcher requires SiLU
    assert p._pinned_ok(desc(64, 64, 1, ACT_NONE), 132)                 # ... tile 132 takes it
    assert p._pinned_ok(desc(64, 128, 2, ACT_SILU), 134) and not p._pinned_ok(desc(64, 128, 2, ACT_NONE), 134)
    assert p._pinned_ok(desc(64, 64, 1, ACT_SILU), 137) and not p._pinned_ok(desc(64, 64, 1, ACT_SILU), 138)   # 138 is the stride-2 form
    assert p._pinned_ok(desc(128, 128, 2, ACT_SILU), 135) and not p._pinned_ok(desc(128, 128, 2, ACT_NONE), 135)
    assert p._pinned_ok(desc(128, 128, 1, ACT_NONE), 143)               # row-transposed-store forms of the general tiles: no such preconditions
    p.res3x3, p.rw2 = 0, 0                                              # YOLORT_AMD_RES3X3=0 / YOLORT_AMD_RW2=0 win over the table
    assert not p._pinned_ok(desc(64, 64, 1, ACT_SILU), 133) and not p._pinned_ok(desc(64, 64, 1, ACT_SILU), 132) and not p._pinned_ok(desc(64, 128, 2, ACT_SILU), 134)


def test_a_rejected_pinned_tile_does_not_come_back_through_the_table_fallback(monkeypatch):
    """Plan.conv's last branch (`elif self.use_tile_table: d.tile = pinned`) re-assigned a table entry >= 132 that _pinned_ok had just rejected.  The concrete
    case: yolov5s r3.1, fp16, bs 32, 640 x 640 -- body.3 = Conv(64, 128, 3, 2) runs with YMI_ACT_NONE (its Hardswish is a launch of its own) on a key the table pins
    to tile 138 (SiLU only): the launch must take the library heuristic (tile 0) instead.  Driven through the whole rule chain of Plan.conv, not _pinned_ok alone."""
    import ctypes as C
    from yolort_amd import _lib, engine
    from yolort_amd._lib import ACT_NONE, ACT_SILU

    class Lib:   # records the descriptor instead of a launch
        def __init__(self):
            self.real, self.tiles = _lib.load(require_gpu=False), []

        def ymi_plan_add_conv(self, h, dref):
            self.tiles.append(int(dref._obj.tile))
            return len(self.tiles) - 1

        def ymi_plan_num_ops(self, h):
            return len(self.tiles)

        def ymi_conv_build_ktab(self, *a):
            return self.real.ymi_conv_build_ktab(*a)

    def plan():
        p = engine.Plan.__new__(engine.Plan)
        p.lib, p.device, p.dtype, p.handle = Lib(), torch.device("cpu"), torch.float16, C.c_void_p(1)
        p.keep, p.names, p.meta, p.bytes_allocated, p.stream = [], [], [], 0, None
        p.conv_descs, p.io = {}, {}
        p.chain_1x1, p.chain_cv3, p.use_v1, p.fuse_c3, p.c3_tile_on, p.chain_next = True, False, False, True, True, False
        p.autotune, p.use_tile_table, p.fp32, p.fuse_stem = False, True, False, False
        p.res3x3, p.rw2, p.rw3, p.rs = 2, 1, False, False
        return p

    w = torch.randn(128, 64, 3, 3)
    seen = {}
    for act, name in ((ACT_SILU, "silu"), (ACT_NONE, "none")):
        p = plan()
        x = p.alloc(32, 160, 160, 64)
        pc = engine.PackedConv(w, None, None, torch.float16, torch.device("cpu"))
        y = p.alloc(32, 80, 80, 128)
        key = engine.tile_key_str((32, 160, 160, 64, 128, 3, 3, (2, 2), (1, 1), 64, 128, _lib.dtype_code(torch.float16), False, 0, False, 0), torch.float16)
        monkeypatch.setattr(engine, "_TILE_TABLE", {key: 138})
        p.conv(x, pc, 2, 1, act, out=y, name="body.3")
        seen[name] = p.lib.tiles[-1]
        p.handle = None
    assert seen["silu"] == 138          # the entry applies to the SiLU launch it was tuned for
    assert seen["none"] == 0, seen      # ... and is NOT forced onto the activation-free launch (conv3x3_rs_launch would refuse it: YMI_EINVAL at plan build)


@pytest.mark.parametrize("walk", ["python", "c"])
def test_weights_signature_sees_every_way_a_module_tree_can_change(walk, monkeypatch):
    """(both forms of the walk: the interpreter-level one and torch_ext/sig_ext.cpp, skipped when `_ymi_sig.so` is not built)
    hipmodule.weights_signature is the plan cache's key: it re-reads the live `_parameters` / `_buffers` / `_modules` dicts on every call (the round-4
    form cached the tensor objects behind process-wide registration hooks and missed `del`, `= None` and `_apply` with overwrite-on-conversion)"""
    import copy

    from torch import nn

    from yolort_amd import hipmodule
    from yolort_amd.hipmodule import weights_signature
    from yolort_amd.models import yolo

    if walk == "python":
        monkeypatch.setattr(hipmodule, "_SIG_EXT", None)
    elif hipmodule._SIG_EXT is None:
        pytest.skip("yolort_amd/lib/_ymi_sig.so is not built (python -m yolort_amd.torch_ext)")
    base = yolo.yolov5_darknet_pan_n_r60().eval()
    s0 = weights_signature(base)
    assert weights_signature(base) == s0 and weights_signature(copy.deepcopy(base)) != s0    # stable; another set of tensors is another key
    import torch.nn.modules.module as M
    assert not M._global_parameter_registration_hooks and not M._global_buffer_registration_hooks and not M._global_module_registration_hooks   # importing the package installs no process-wide hooks

    def changed(mutate):
        m = copy.deepcopy(base)
        a = weights_signature(m)
        mutate(m)
        return weights_signature(m) != a

    assert changed(lambda m: m.head.head.__delitem__(2))                                      # a sub-module deleted (__delattr__ fires no registration hook)
    assert changed(lambda m: setattr(m.head.head[0], "bias", None))                           # register_parameter(None) fires none either
    def convert(m):
        torch.__future__.set_overwrite_module_params_on_conversion(True)
        try:
            m.double()                                                                        # _apply writes NEW Parameter objects straight into _parameters
        finally:
            torch.__future__.set_overwrite_module_params_on_conversion(False)
    assert changed(convert)
    def inplace(m):
        with torch.no_grad():
            m.head.head[0].weight.add_(1.0)
    assert changed(inplace)                                                                   # _version
    assert changed(lambda m: m.half())                                                        # same Parameter objects, new storage: data_ptr
    assert changed(lambda m: m.backbone.body["0"].act.register_buffer("k", torch.zeros(1)))   # a buffer on a module that had no tensors
    assert changed(lambda m: m.backbone.body["0"].act.add_module("extra", nn.Conv2d(1, 1, 1)))  # a child under a former leaf
    assert changed(lambda m: m.head.head.__setitem__(1, nn.Conv2d(256, 255, 1)))              # a replaced sub-module

    def delete_then_recreate_the_last_child(m):                                               # the freed address is what CPython hands the new instance of the same class;
        last = list(m.head.head._modules)[-1]                                                 # the cache keeps the old child alive, so the identities differ
        old = m.head.head._modules[last]
        spec = (old.in_channels, old.out_channels, old.kernel_size)
        del m.head.head._modules[last]
        del old
        m.head.head._modules[last] = nn.Conv2d(*spec)
    for _ in range(5):
        assert changed(delete_then_recreate_the_last_child)
