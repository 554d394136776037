"""Generates tests/golden/*.npz|json by running the UNMODIFIED reference (read-only import from
/root/reference, torchvision stand-in from oracle/_tv_compat.py).  Build-container only.

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.reference_loader import load_reference  # noqa: E402
from workloads.synth import synth_images, synth_weights  # noqa: E402

yolort = load_reference()
from yolort.models import YOLOv5  # noqa: E402
from yolort.models.anchor_utils import AnchorGenerator  # noqa: E402
from yolort.models.box_head import _concat_pred_logits, _decode_pred_logits  # noqa: E402
from yolort.models.transform import YOLOTransform, scale_coords  # noqa: E402

LETTERBOX_SHAPES = [(1080, 810), (480, 640), (720, 1280), (1080, 1920), (375, 500), (427, 640), (333, 500),
                    (1281, 1279), (641, 480), (100, 37)]


def letterbox_golden():
    out = {"cases": []}
    for S in (640, 1280):
        for stride in (32, 64):
            t = YOLOTransform(S, S, size_divisible=stride)
            for (h, w) in LETTERBOX_SHAPES:
                img = synth_images(1, h, w, seed=h * 7 + w)[0]
                nt, _ = t([img])
                hh, ww = nt.image_sizes[0]
                canvas = tuple(nt.tensors.shape[-2:])
                x = nt.tensors[0]
                # locate top-left of the placed image via the reference's own rule is implicit; store checksums
                out["cases"].append({"S": S, "stride": stride, "hw": [h, w], "resized": [int(hh), int(ww)], "canvas": list(canvas),
                                     "sum": float(x.double().sum()), "sample": [float(v) for v in x[:, ::max(1, canvas[0] // 7), ::max(1, canvas[1] // 5)].flatten()[:64]]})
    # mixed batch + fixed shape
    for fixed in (None, (640, 640)):
        t = YOLOTransform(640, 640, fixed_shape=fixed)
        imgs = [synth_images(1, h, w, seed=h + w)[0] for (h, w) in [(1080, 810), (480, 640), (720, 1280)]]
        nt, _ = t(imgs)
        out["cases"].append({"mixed": True, "fixed": list(fixed) if fixed else None, "canvas": list(nt.tensors.shape),
                             "image_sizes": [list(map(int, s)) for s in nt.image_sizes], "sum": float(nt.tensors.double().sum())})
    t = YOLOTransform(640, 640, fixed_shape=(640, 640))
    nt, _ = t([synth_images(1, 427, 640, seed=3)[0]])
    rows = (nt.tensors[0, 0, :, 320] != 114 / 255).nonzero().flatten()
    out["fixed_427"] = {"first_row": int(rows[0]), "last_row": int(rows[-1])}
    out["fill"] = float(torch.tensor(114 / 255, dtype=torch.float32))
    b = scale_coords(torch.tensor([[100.0, 100.0, 200.0, 200.0]]), torch.tensor([640, 480]), (1080, 810))
    out["scale_coords"] = [float(v) for v in b.flatten()]
    b = scale_coords(torch.tensor([[10.5, 20.25, 300.75, 333.0]]), torch.tensor([384, 640]), (720, 1280))
    out["scale_coords2"] = [float(v) for v in b.flatten()]
    json.dump(out, open(os.path.join(HERE, "letterbox.json"), "w"))


def anchors_decode_golden():
    # reference test/test_models_anchor_utils.py:14-30 setup
    ag = AnchorGenerator([4], [[6, 14]])
    grids, shifts = ag([torch.rand(1, 3, 2, 2)])
    # SURVEY.md Appendix F setup
    ag2 = AnchorGenerator([8], [[10, 13, 16, 30, 33, 23]])
    ho = [torch.full((1, 3, 2, 3, 85), float(np.log(3.0)))]
    g2, s2 = ag2([torch.rand(1, 4, 2, 3)])
    pred = _concat_pred_logits(ho, g2, s2, torch.tensor([8.0]))
    boxes, scores = _decode_pred_logits(pred[0])
    # random head outputs through the reference decode (reference test pattern test/test_models.py:153-165)
    torch.manual_seed(7)
    ho3 = [torch.randn(2, 3, 4, 5, 85) * 2, torch.randn(2, 3, 2, 3, 85) * 2]
    ag3 = AnchorGenerator([8, 16], [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119]])
    g3, s3 = ag3([torch.rand(1, 1, 4, 5), torch.rand(1, 1, 2, 3)])
    pred3 = _concat_pred_logits(ho3, g3, s3, torch.tensor([8.0, 16.0]))
    np.savez(os.path.join(HERE, "anchors_decode.npz"), grids=grids[0].numpy(), shifts=shifts[0].numpy(),
             kat_pred=pred.numpy(), kat_boxes=boxes.numpy(), kat_scores=scores.numpy(),
             rnd_ho0=ho3[0].numpy(), rnd_ho1=ho3[1].numpy(), rnd_pred=pred3.numpy())


def e2e_golden(arch="yolov5_darknet_pan_n_r60", tag="n", sizes=((160, 120), (96, 160), (128, 128)), S=160, head_gain=1.0, thr=0.1):
    model = YOLOv5(arch=arch, size=(S, S), score_thresh=thr, nms_thresh=0.45)
    sd = synth_weights(model.state_dict(), arch, seed=0, head_gain=head_gain)
    model.load_state_dict(sd)
    model.eval()
    imgs = [synth_images(1, h, w, seed=11 + i)[0] for i, (h, w) in enumerate(sizes)]
    stages = {}
    model.model.backbone.register_forward_hook(lambda m, i, o: stages.__setitem__("features", o))
    model.model.head.register_forward_hook(lambda m, i, o: stages.__setitem__("head", o))
    model.transform.register_forward_hook(lambda m, i, o: stages.__setitem__("batch", o[0].tensors))
    with torch.no_grad():
        dets = model.predict(imgs)
    out = {"sizes": np.asarray(sizes), "S": S, "head_gain": head_gain, "thr": thr, "batch_shape": np.asarray(stages["batch"].shape)}
    out["batch_sum"] = float(stages["batch"].double().sum())
    for i, f in enumerate(stages["features"]):
        out[f"feat{i}"] = f.numpy().astype(np.float32)
    for i, h in enumerate(stages["head"]):
        out[f"head{i}"] = h.numpy().astype(np.float32)
    for i, d in enumerate(dets):
        out[f"det{i}_boxes"] = d["boxes"].numpy()
        out[f"det{i}_scores"] = d["scores"].numpy()
        out[f"det{i}_labels"] = d["labels"].numpy()
        print(tag, "image", i, "detections", len(d["scores"]), "labels", len(set(d["labels"].tolist())))
    np.savez_compressed(os.path.join(HERE, f"e2e_{tag}.npz"), **out)


# ---------------------------------------------------------------------------------------------------------------------------
# round 3: goldens that can carry a tolerance.  (1) the CONDITIONED workload (workloads/synth.py COND_*): reference detections,
# the reference's own fp32-vs-fp64 reproducibility, and the tolerance a 16-bit evaluation can be held to, measured with jittered
# storage emulations BEFORE it is written into a GPU test.  (2) the reference's asset photos through `predict(path)`.
# ---------------------------------------------------------------------------------------------------------------------------
COND_TAGS = {"yolov5_darknet_pan_n_r60": "n", "yolov5_darknet_pan_s_r60": "s", "yolov5_darknet_pan_m_r60": "m", "yolov5_darknet_pan_l6_r60": "l6",
             "yolov5_darknet_pan_s_r40": "s_r40", "yolov5_darknet_pan_s_r31": "s_r31"}


def _np_dets(dets):
    return [{k: v.detach().cpu().numpy() for k, v in d.items()} for d in dets]


def _reference_model(arch, S, thr, sd, dtype=torch.float32):
    kw = dict(size_divisible=64) if arch.endswith("6_r60") else {}
    model = YOLOv5(arch=arch, size=(S, S), score_thresh=thr, nms_thresh=0.45, **kw)
    model.load_state_dict(sd)
    return model.eval().to(dtype)


def _emulated(imgs, sd, S, div, thr, dtype, jitter_seed=None):
    from oracle import yolov5_oracle as O
    O.EMULATE.dtype = dtype
    O.EMULATE.jitter = None if jitter_seed is None else torch.Generator().manual_seed(jitter_seed)
    try:
        with torch.no_grad():
            return _np_dets(O.yolov5_forward(imgs, sd, size=(S, S), size_divisible=div, score_thresh=thr))
    finally:
        O.EMULATE.dtype, O.EMULATE.jitter = None, None


def _tolerance_of(ref, runs, thr):
    """the loosest (min IoU, max |dscore|) over the runs, pairing with generous bounds; unpaired = detections no generous pairing explains"""
    sys.path.insert(0, ROOT)
    import bench
    worst = {"min_iou": 1.0, "max_dscore": 0.0, "unexplained": 0, "at_cut": 0, "paired": 10 ** 9}
    for got in runs:
        c = bench.direct_checks(ref, got, thr, score_eps=0.1, iou_min=0.5)
        worst["min_iou"] = min(worst["min_iou"], c["min_iou"])
        worst["max_dscore"] = max(worst["max_dscore"], c["max_dscore"])
        worst["unexplained"] = max(worst["unexplained"], c["unexplained"])
        worst["at_cut"] = max(worst["at_cut"], c["at_cut"])
        worst["paired"] = min(worst["paired"], c["paired"])
    return worst


def cond_evaluate(arch, seed, thr=0.25, jitters=(None, 1, 2, 3), verbose=True):
    """reference detections of the conditioned workload + how reproducible they are (fp64, jittered fp16 / bf16 storage)"""
    import bench
    from oracle import yolov5_oracle as O
    from workloads.synth import COND_SIZE, cond_images, conditioned_weights
    S = COND_SIZE[arch]
    div = 64 if arch.endswith("6_r60") else 32
    kw = dict(size_divisible=64) if div == 64 else {}
    tmpl = YOLOv5(arch=arch, size=(S, S), **kw).state_dict()
    sd = conditioned_weights(tmpl, arch, seed)
    imgs = cond_images(arch, seed)
    with torch.no_grad():
        ref = _np_dets(_reference_model(arch, S, thr, sd).predict(imgs))                    # the UNMODIFIED reference, fp32
        ref64 = _np_dets(_reference_model(arch, S, thr, sd, torch.float64).predict([im.double() for im in imgs]))   # ... and in float64
        ora = _np_dets(O.yolov5_forward(imgs, sd, size=(S, S), size_divisible=div, score_thresh=thr))
    for r in ref64:
        r["boxes"], r["scores"] = r["boxes"].astype(np.float32), r["scores"].astype(np.float32)
    c64 = bench.direct_checks(ref, ref64, thr, score_eps=1e-4, iou_min=1 - 1e-3)
    cor = bench.direct_checks(ref, ora, thr, score_eps=1e-4, iou_min=1 - 1e-3)
    # margins of the discrete decisions of the reference run: score gaps between consecutive detections of an image (label-sequence
    # equality needs |dscore| below half of it) and the distance of every detection from the threshold
    gap, thr_margin = 1.0, 1.0
    for r in ref:
        s = r["scores"]
        if len(s) > 1:
            gap = min(gap, float(np.min(-np.diff(s))))
        if len(s):
            thr_margin = min(thr_margin, float(s.min() - thr))
    # the eight emulated 16-bit evaluations are three quarters of a seed's cost (yolov5l6: ~35 of 45 minutes): a seed the exact criteria already reject skips them
    n_img = len(ref)
    exact_ok = (all(c["unexplained"] == 0 and c["at_cut"] == 0 and c["images_labels_equal"] == n_img for c in (c64, cor)) and gap >= 1e-4 and thr_margin >= 5e-5
                and max(len(r["scores"]) for r in ref) <= 150)
    tol = None
    if exact_ok:
        tol = {}
        for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            tol[name] = _tolerance_of(ref, [_emulated(imgs, sd, S, div, thr, dt, j) for j in jitters], thr)
    out = {"arch": arch, "seed": seed, "S": S, "thr": thr, "dets": [len(r["scores"]) for r in ref], "fp64": c64, "oracle": cor, "min_score_gap": gap,
           "thr_margin": thr_margin, "tol": tol}
    if verbose:
        print(json.dumps(out), flush=True)
    return out, ref


def cond_ok(ev):
    """a seed is usable when the reference reproduces itself exactly in fp64 (every detection paired at 1 - 1e-3, same label sequence), the
    restatement agrees, the discrete margins are wide against fp32 noise, and no 16-bit rounding history loses a detection outright"""
    n = len(ev["dets"])
    if ev["tol"] is None:   # rejected by the exact criteria before the 16-bit emulation
        return False
    ok64 = all(ev[k]["unexplained"] == 0 and ev[k]["at_cut"] == 0 and ev[k]["images_labels_equal"] == n for k in ("fp64", "oracle"))
    low = "bf16" if ev["arch"].endswith("_m_r60") else "fp16"   # the 16-bit type the architecture's BASELINE config runs in
    if low == "bf16":
        # yolov5m at 1280 x 1280 in bf16 (8 mantissa bits): the 16-bit tolerance is loose by nature; the seed is chosen for the EXACT part (fp64
        # reproducibility, margins) and whatever a bf16 evaluation then meets is recorded and stated as such in tests/test_golden_gpu.py
        return ok64 and max(ev["dets"]) >= 10 and max(ev["dets"]) <= 150 and ev["min_score_gap"] >= 1e-4 and ev["thr_margin"] >= 5e-5
    return (ok64 and sum(1 for d in ev["dets"] if d >= 2) >= 2 and max(ev["dets"]) <= 150 and ev["min_score_gap"] >= 1e-4 and ev["thr_margin"] >= 5e-5
            and ev["tol"][low]["unexplained"] == 0 and ev["tol"][low]["min_iou"] >= 0.97)


def cond_golden(arch, seeds=range(0, 40)):
    """searches the seed range for a usable conditioned workload, commits its calibration file and the reference's detections"""
    import subprocess
    from workloads.synth import cond_bn_path
    tag = COND_TAGS[arch]
    for seed in seeds:
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_synth_bn.py"), "--cond", f"--seed={seed}", arch], check=True, capture_output=True)
        ev, ref = cond_evaluate(arch, seed)
        if cond_ok(ev):
            out = {"meta": json.dumps(ev)}
            for i, r in enumerate(ref):
                for k in ("boxes", "scores", "labels"):
                    out[f"det{i}_{k}"] = r[k]
            np.savez_compressed(os.path.join(HERE, f"cond_{tag}.npz"), **out)
            print("conditioned golden", tag, "seed", seed, "dets", ev["dets"], "tolerances", ev["tol"])
            return seed
        os.remove(cond_bn_path(arch, seed))
    raise RuntimeError(f"no usable seed for {arch} in {list(seeds)}")


ASSETS = "/root/reference/test/assets"


def photo_pngs():
    """the reference's asset photos, decoded (PIL = libjpeg-turbo, what torchvision.io.read_image wraps) and committed losslessly: the GPU box has
    no /root/reference, `predict(path)` there reads these PNGs and gets the identical uint8 arrays"""
    from PIL import Image
    for name in ("bus", "zidane"):
        a = np.asarray(Image.open(os.path.join(ASSETS, name + ".jpg")).convert("RGB"))
        Image.fromarray(a).save(os.path.join(HERE, name + ".png"), optimize=True)
        b = np.asarray(Image.open(os.path.join(HERE, name + ".png")).convert("RGB"))
        assert np.array_equal(a, b)
        print(name, a.shape, os.path.getsize(os.path.join(HERE, name + ".png")) // 1024, "KB")


def photo_evaluate(arch, seed, thr=0.25, jitters=(None, 1, 2)):
    """`model.predict([bus.jpg, zidane.jpg])` of the UNMODIFIED reference (its own default_loader on the JPEG files) with the conditioned weights
    calibrated on the photos"""
    import bench
    from oracle import yolov5_oracle as O
    from oracle.make_synth_bn import photo_images
    from workloads.synth import COND_SIZE, conditioned_weights
    S = COND_SIZE[arch]
    div = 64 if arch.endswith("6_r60") else 32
    kw = dict(size_divisible=64) if div == 64 else {}
    sd = conditioned_weights(YOLOv5(arch=arch, size=(S, S), **kw).state_dict(), arch, seed, variant="photo")
    paths = [os.path.join(ASSETS, n + ".jpg") for n in ("bus", "zidane")]
    imgs = photo_images()
    with torch.no_grad():
        ref = _np_dets(_reference_model(arch, S, thr, sd).predict(paths))
        one = [_np_dets(_reference_model(arch, S, thr, sd).predict(pth))[0] for pth in paths]     # single paths: other canvases (the batch maximum differs)
        ref64 = _np_dets(_reference_model(arch, S, thr, sd, torch.float64).predict([im.double() for im in imgs]))
        ora = _np_dets(O.yolov5_forward(imgs, sd, size=(S, S), size_divisible=div, score_thresh=thr))
    for r in ref64:
        r["boxes"], r["scores"] = r["boxes"].astype(np.float32), r["scores"].astype(np.float32)
    c64 = bench.direct_checks(ref, ref64, thr, score_eps=1e-4, iou_min=1 - 1e-3)
    cor = bench.direct_checks(ref, ora, thr, score_eps=1e-4, iou_min=1 - 1e-3)
    gap, thr_margin = 1.0, 1.0
    for r in ref + one:
        s = r["scores"]
        if len(s) > 1:
            gap = min(gap, float(np.min(-np.diff(s))))
        if len(s):
            thr_margin = min(thr_margin, float(s.min() - thr))
    tol = {}
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        tol[name] = _tolerance_of(ref, [_emulated(imgs, sd, S, div, thr, dt, j) for j in jitters], thr)
    ev = {"arch": arch, "seed": seed, "S": S, "thr": thr, "dets": [len(r["scores"]) for r in ref], "dets_single": [len(r["scores"]) for r in one], "fp64": c64,
          "oracle": cor, "min_score_gap": gap, "thr_margin": thr_margin, "tol": tol}
    print(json.dumps(ev), flush=True)
    return ev, ref, one


def photo_golden(arch, seeds=range(0, 30)):
    import subprocess
    from workloads.synth import cond_bn_path
    tag = COND_TAGS[arch]
    for seed in seeds:
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_synth_bn.py"), "--cond", "--photo", f"--seed={seed}", arch], check=True, capture_output=True)
        ev, ref, one = photo_evaluate(arch, seed)
        n = len(ev["dets"])
        ok = (all(ev[k]["unexplained"] == 0 and ev[k]["at_cut"] == 0 and ev[k]["images_labels_equal"] == n for k in ("fp64", "oracle"))
              and min(ev["dets"]) >= 2 and max(ev["dets"]) <= 100 and max(ev["dets_single"]) >= 2 and ev["min_score_gap"] >= 1e-4 and ev["thr_margin"] >= 5e-5)
        if ok:
            out = {"meta": json.dumps(ev)}
            for i, r in enumerate(ref):
                for k in ("boxes", "scores", "labels"):
                    out[f"det{i}_{k}"] = r[k]
            for j, o in enumerate(one):   # `predict(path)` of each photo alone
                for k in ("boxes", "scores", "labels"):
                    out[f"single{j}_{k}"] = o[k]
            np.savez_compressed(os.path.join(HERE, f"photo_{tag}.npz"), **out)
            print("photo golden", tag, "seed", seed, "dets", ev["dets"], "single", ev["dets_single"], "tolerances", ev["tol"])
            return seed
        os.remove(cond_bn_path(arch, seed, "photo"))
    raise RuntimeError(f"no usable seed for {arch} in {list(seeds)}")


# ---------------------------------------------------------------------------------------------------------------------------
# round 4: the reference's OWN 16-bit behaviour (VERDICT r3 "What's missing" 3).  The unmodified reference with `.half()` / `.bfloat16()`
# parameters on the same inputs (yolov5.py:202-216; everything after the convolutions -- sigmoid, grid arithmetic, box conversion, NMS,
# rescale -- then also runs in that type, as it does on the reference's own GPU path).  Its distance from its own fp32 detections is the
# like-for-like yardstick for the HIP 16-bit path's tolerance: tests/test_golden_gpu.py asserts the HIP path is no further away.
# ---------------------------------------------------------------------------------------------------------------------------
def _np_dets32(dets):
    return [{k: (v.detach().float().cpu().numpy() if v.is_floating_point() else v.detach().cpu().numpy()) for k, v in d.items()} for d in dets]


def _band(ref, got, thr):
    """distance of a 16-bit evaluation from the fp32 detections: generous pairing (same label, IoU >= 0.5, |dscore| <= 0.1), then the worst pair"""
    import bench
    c = bench.direct_checks(ref, got, thr, score_eps=0.1, iou_min=0.5)
    return {"ref_dets": c["ref_dets"], "dets": c["hip_dets"], "paired": c["paired"], "min_iou": c["min_iou"], "iou_deficit": round(1.0 - c["min_iou"], 6), "max_dscore": c["max_dscore"],
            "unpaired_ref": c["ref_dets"] - c["paired"], "unpaired_got": c["hip_dets"] - c["paired"]}


def ref16_golden(kind, tag):
    from oracle.make_synth_bn import photo_images
    from workloads.synth import cond_images, conditioned_weights, spread_images
    by_tag = {v: k for k, v in COND_TAGS.items()}
    arch = by_tag[tag] if tag in by_tag else by_tag[tag.split("_")[0]]   # (tag "s_s3": the extra seeds of spread_more; "s_r40" / "s_r31": the legacy releases)
    z = np.load(os.path.join(HERE, f"{kind}_{tag}.npz"))
    meta = json.loads(str(z["meta"]))
    S, thr, seed = meta["S"], meta["thr"], meta["seed"]
    kw = dict(size_divisible=64) if arch.endswith("6_r60") else {}
    variant = meta.get("variant", "photo" if kind == "photo" else "cond")
    sd = conditioned_weights(YOLOv5(arch=arch, size=(S, S), **kw).state_dict(), arch, seed, variant=variant)
    imgs = photo_images() if kind == "photo" else (cond_images(arch, seed) if variant in ("cond", "lin") else spread_images(arch, seed))
    ref = [{k: z[f"det{i}_{k}"] for k in ("boxes", "scores", "labels")} for i in range(len(meta["dets"]))]
    out = {}
    info = {"arch": arch, "seed": seed, "S": S, "thr": thr, "kind": kind, "what": "detections of the UNMODIFIED reference with .half() / .bfloat16() parameters and inputs (CPU)"}
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        with torch.no_grad():
            got = _np_dets32(_reference_model(arch, S, thr, sd, dt).predict([im.to(dt) for im in imgs]))
        info[name] = _band(ref, got, thr)
        for i, g in enumerate(got):
            for k in ("boxes", "scores", "labels"):
                out[f"{name}_det{i}_{k}"] = g[k]
        print(kind, tag, name, info[name], flush=True)
    out["meta"] = json.dumps(info)
    np.savez_compressed(os.path.join(HERE, f"ref16_{kind}_{tag}.npz"), **out)


# ---------------------------------------------------------------------------------------------------------------------------
# round 4: the SPREAD workload (workloads/synth.py SPREAD_*): reference scores from the threshold up to ~0.9 and the score threshold
# placed in a GAP of the reference's own score list, so that no detection is "near the cut" and a 16-bit evaluation has to reproduce every
# single one (the conditioned workload's scores all lie within a few hundredths of the threshold).
# ---------------------------------------------------------------------------------------------------------------------------
def spread_evaluate(arch, seed, thr_range=(0.3, 0.8), verbose=True, variant="spread", min_dets=40):
    """variant "spread": the spread recipe; variant "cond": the CONDITIONED recipe (round 3) evaluated the same way -- the threshold goes into a gap of the
    reference's score list instead of sitting at 0.25 (yolov5l6: the spread recipe's gain-4 head is not reproducible in fp32 on the P6 network -- its fp64
    run re-decides 30-100 detections, tests/golden/spread_l6_search.txt -- so its reference-made golden is a conditioned one with a gap threshold)"""
    import bench
    from oracle import yolov5_oracle as O
    from workloads.synth import COND_SIZE, cond_images, conditioned_weights, spread_images
    S = COND_SIZE[arch]
    div = 64 if arch.endswith("6_r60") else 32
    kw = dict(size_divisible=64) if div == 64 else {}
    sd = conditioned_weights(YOLOv5(arch=arch, size=(S, S), **kw).state_dict(), arch, seed, variant=variant)
    imgs = spread_images(arch, seed) if variant == "spread" else cond_images(arch, seed)
    with torch.no_grad():
        low = _np_dets(_reference_model(arch, S, 0.15, sd).predict(imgs))     # everything down to 0.15: the score list the threshold is placed in
    pool = np.sort(np.concatenate([d["scores"] for d in low] + [np.asarray([0.15, 1.0], np.float32)]))
    # candidate thresholds: the middle of every gap of the pooled score list inside thr_range that keeps at least 40 detections above it; the widest one wins
    gaps = [(float(b - a), float(0.5 * (a + b))) for a, b in zip(pool[:-1], pool[1:]) if thr_range[0] <= 0.5 * (a + b) <= thr_range[1] and int((pool >= b).sum()) - 1 >= min_dets]
    if not gaps:
        return {"arch": arch, "seed": seed, "dets": [len(d["scores"]) for d in low], "gap": 0.0}, None
    gap, thr = max(gaps)
    thr = round(thr, 4)
    with torch.no_grad():
        ref = _np_dets(_reference_model(arch, S, thr, sd).predict(imgs))
        ref64 = _np_dets(_reference_model(arch, S, thr, sd, torch.float64).predict([im.double() for im in imgs]))
        ora = _np_dets(O.yolov5_forward(imgs, sd, size=(S, S), size_divisible=div, score_thresh=thr))
    for r in ref64:
        r["boxes"], r["scores"] = r["boxes"].astype(np.float32), r["scores"].astype(np.float32)
    c64 = bench.direct_checks(ref, ref64, thr, score_eps=1e-4, iou_min=1 - 1e-3)
    cor = bench.direct_checks(ref, ora, thr, score_eps=1e-4, iou_min=1 - 1e-3)
    sgap = 1.0
    for r in ref:
        if len(r["scores"]) > 1:
            sgap = min(sgap, float(np.min(-np.diff(r["scores"]))))
    allsc = np.concatenate([r["scores"] for r in ref]) if sum(len(r["scores"]) for r in ref) else np.zeros(0, np.float32)
    own = {}
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):   # the reference's own 16-bit evaluation of this workload
        with torch.no_grad():
            own[name] = _band(ref, _np_dets32(_reference_model(arch, S, thr, sd, dt).predict([im.to(dt) for im in imgs])), thr)
    ev = {"arch": arch, "seed": seed, "S": S, "thr": thr, "variant": variant, "dets": [len(r["scores"]) for r in ref], "thr_gap": gap,
          "thr_margin": float(allsc.min() - thr) if len(allsc) else 0.0, "min_score_gap": sgap,
          "score_range": [float(allsc.min()), float(allsc.max())] if len(allsc) else None,
          "score_quartiles": [float(q) for q in np.quantile(allsc, [0.25, 0.5, 0.75])] if len(allsc) else None,
          "fp64": c64, "oracle": cor, "reference_own_16bit": own}
    if verbose:
        print(json.dumps(ev), flush=True)
    return ev, ref


def spread_ok(ev):
    n = len(ev["dets"])
    if ev.get("gap", 1.0) == 0.0 or "fp64" not in ev:
        return False
    low = "bf16" if ev["arch"].endswith("_m_r60") else "fp16"
    own = ev["reference_own_16bit"][low]
    exact = all(ev[k]["unexplained"] == 0 and ev[k]["at_cut"] == 0 and ev[k]["images_labels_equal"] == n for k in ("fp64", "oracle"))
    # usable: exact in fp32 / fp64 / the restatement (every detection, identical label sequences); detections on at least two images; consecutive scores further
    # apart than 5e-5 (an fp32 summation order moves a score by ~2e-6); scores up to 0.75 at least; and the threshold's margin (half the gap) no smaller than
    # the score error of the REFERENCE'S OWN evaluation in the architecture's 16-bit type (fp16; bf16 for yolov5m: half of it -- its own bf16 run moves scores by 0.05-0.1)
    need = own["max_dscore"] * (0.5 if low == "bf16" else 1.0)
    return (exact and sum(1 for d in ev["dets"] if d >= 3) >= 2 and 40 <= sum(ev["dets"]) <= 400 and ev["min_score_gap"] >= 5e-5
            and ev["score_range"][1] >= 0.75 and 0.5 * ev["thr_gap"] >= need)


def spread_golden(arch, seeds=range(0, 40), force=False):
    """`force`: commit the given seed whatever the margins are (they are recorded in the golden's meta and the GPU test states what it can assert): yolov5m in bf16
    -- the reference's OWN bfloat16 run moves scores by 0.05-0.1 and re-decides a fifth to three quarters of the detections, no threshold gap of a hundred-detection
    workload is that wide (tests/golden/spread_m_search.txt) -- and yolov5l6, whose evaluations cost minutes each (a bounded search, then the best seed)"""
    import subprocess
    from workloads.synth import cond_bn_path
    tag = COND_TAGS[arch]
    for seed in seeds:
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_synth_bn.py"), "--cond", "--spread", f"--seed={seed}", arch], check=True, capture_output=True)
        ev, ref = spread_evaluate(arch, seed)
        if spread_ok(ev) or (force and ref is not None):
            ev["accepted_by"] = "criteria (spread_ok)" if spread_ok(ev) else "forced: best seed of a bounded search, margins as recorded"
            out = {"meta": json.dumps(ev)}
            for i, r in enumerate(ref):
                for k in ("boxes", "scores", "labels"):
                    out[f"det{i}_{k}"] = r[k]
            np.savez_compressed(os.path.join(HERE, f"spread_{tag}.npz"), **out)
            print("spread golden", tag, "seed", seed, "thr", ev["thr"], "dets", ev["dets"], "reference's own 16-bit", ev["reference_own_16bit"])
            ref16_golden("spread", tag)
            return seed
        os.remove(cond_bn_path(arch, seed, "spread"))
    raise RuntimeError(f"no usable seed for {arch} in {list(seeds)}")


def spread_more(arch, seeds, want=8):
    """VERDICT r3 weak 3 (one seed of four images per architecture is thin): further seeds of the spread workload, each with its own weights, images and gap threshold,
    accepted when the reference is EXACTLY reproducible on it (fp32 = fp64 = the restatement: every detection, identical label sequences), consecutive scores are
    further apart than 5e-5 and at least two images carry detections -- what the fp32-mode test and the "no further than the reference's own 16-bit run" test need;
    meta["strict_16bit"] says whether the seed also meets the FIRST seed's margin criterion (spread_ok: half the threshold gap >= the reference's own 16-bit score
    error), which the "pairs every detection" test needs.  Search record of 16 seeds of yolov5s: all 16 exact, 10 with the score spacing, 1 with the margin
    (tests/golden/spread_s_more_search.txt).  Files spread_<tag>_s<seed>.npz + ref16_spread_<tag>_s<seed>.npz.  Stops after `want` accepted seeds."""
    import subprocess
    from workloads.synth import cond_bn_path
    tag = COND_TAGS[arch]
    first = json.loads(str(np.load(os.path.join(HERE, f"spread_{tag}.npz"))["meta"]))["seed"]
    got = []
    for seed in seeds:
        if seed == first or len(got) >= want:
            continue
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_synth_bn.py"), "--cond", "--spread", f"--seed={seed}", arch], check=True, capture_output=True)
        ev, ref = spread_evaluate(arch, seed)
        n_img = len(ev.get("dets", []))
        exact = "fp64" in ev and all(ev[k]["unexplained"] == 0 and ev[k]["at_cut"] == 0 and ev[k]["images_labels_equal"] == n_img for k in ("fp64", "oracle"))
        if ref is not None and exact and ev["min_score_gap"] >= 5e-5 and sum(1 for d in ev["dets"] if d >= 3) >= 2 and 40 <= sum(ev["dets"]) <= 400:
            ev["strict_16bit"] = bool(spread_ok(ev))
            ev["accepted_by"] = "exact fp32 / fp64 / restatement agreement, score spacing >= 5e-5" + (" and the first seed's margin criterion (spread_ok)" if ev["strict_16bit"] else "")
            out = {"meta": json.dumps(ev)}
            for i, r in enumerate(ref):
                for k in ("boxes", "scores", "labels"):
                    out[f"det{i}_{k}"] = r[k]
            np.savez_compressed(os.path.join(HERE, f"spread_{tag}_s{seed}.npz"), **out)
            ref16_golden("spread", f"{tag}_s{seed}")
            got.append(seed)
            print("spread golden", tag, "seed", seed, "thr", ev["thr"], "dets", ev["dets"], flush=True)
        else:
            os.remove(cond_bn_path(arch, seed, "spread"))
    return got


def cond_gap_golden(arch, seeds, force_last=True, variant="cond", min_dets=12):
    """the conditioned golden with its threshold in a gap of the reference's score list (`cond_<tag>.npz`, meta["thr"] instead of 0.25): the first seed whose fp32 / fp64 /
    restatement runs agree exactly is committed; with `force_last` the last seed tried is committed whatever its margins are (recorded in the meta: VERDICT r3 item 2 --
    a bounded search, then the best seed with its margins)"""
    import subprocess
    from workloads.synth import cond_bn_path
    tag = COND_TAGS[arch]
    seeds = list(seeds)
    for n_, seed in enumerate(seeds):
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_synth_bn.py"), "--cond", f"--seed={seed}", arch] + (["--lin"] if variant == "lin" else []), check=True, capture_output=True)
        ev, ref = spread_evaluate(arch, seed, thr_range=(0.22, 0.5), variant=variant, min_dets=min_dets)
        n = len(ev.get("dets", []))
        exact = "fp64" in ev and all(ev[k]["unexplained"] == 0 and ev[k]["at_cut"] == 0 and ev[k]["images_labels_equal"] == n for k in ("fp64", "oracle"))
        if ref is not None and (exact or (force_last and n_ == len(seeds) - 1)):
            ev["accepted_by"] = "exact fp32 / fp64 / restatement agreement" if exact else "forced: last seed of a bounded search, margins as recorded"
            out = {"meta": json.dumps(ev)}
            for i, r in enumerate(ref):
                for k in ("boxes", "scores", "labels"):
                    out[f"det{i}_{k}"] = r[k]
            np.savez_compressed(os.path.join(HERE, f"{variant}_{tag}.npz"), **out)
            print(variant, "gap golden", tag, "seed", seed, "thr", ev["thr"], "dets", ev["dets"], ev["accepted_by"], flush=True)
            ref16_golden(variant, tag)
            return seed
        os.remove(cond_bn_path(arch, seed, variant))
    raise RuntimeError(f"no usable seed for {arch} in {seeds}")


# ---------------------------------------------------------------------------------------------------------------------------
# round 6: the TIMED workload as a golden.  bench.py's headline configuration -- yolov5s, `synth_weights(seed 0, head_gain 0.4)`, the first
# images of `synth_images(32, 640, 640, seed=1)`, score_thresh 0.25 -- through the UNMODIFIED reference: bench.py's parity block and the tests score the HIP path of
# the benchmark against these detections instead of the oracle's.
# ---------------------------------------------------------------------------------------------------------------------------
BENCH_CONFIGS = {"c2": dict(arch="yolov5_darknet_pan_s_r60", size=640, batch=32, thr=0.25, head_gain=0.4, images=8)}


def bench_golden(config="c2"):
    c = BENCH_CONFIGS[config]
    model = YOLOv5(arch=c["arch"], size=(c["size"], c["size"]), score_thresh=c["thr"], nms_thresh=0.45)
    model.load_state_dict(synth_weights(model.state_dict(), c["arch"], seed=0, head_gain=c["head_gain"]))
    model.eval()
    imgs = list(synth_images(c["batch"], c["size"], c["size"], seed=1))[: c["images"]]
    with torch.no_grad():
        dets = _np_dets(model.predict(imgs))
        d64 = _np_dets(model.double().predict([im.double() for im in imgs]))   # the reference against its own float64 evaluation: what fp32 summation order alone moves
    out = {"meta": json.dumps({"config": config, **c, "seed": 0, "image_seed": 1, "dets": [int(len(d["scores"])) for d in dets],
                               "what": "detections of the UNMODIFIED reference (yolort.models.YOLOv5.predict, fp32 CPU) on bench.py's timed workload"})}
    for i, (d, e) in enumerate(zip(dets, d64)):
        for k in ("boxes", "scores", "labels"):
            out[f"det{i}_{k}"] = d[k]
            out[f"f64_{i}_{k}"] = e[k].astype(d[k].dtype) if k != "labels" else e[k]
        print(config, "image", i, "detections", len(d["scores"]), "(float64 evaluation:", len(e["scores"]), ")")
    np.savez_compressed(os.path.join(HERE, f"bench_{config}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bench":   # usage: bench [config]
        bench_golden(*(sys.argv[2:3] or ["c2"]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cond-gap":   # usage: cond-gap arch seed [seed ...]
        cond_gap_golden(sys.argv[2], [int(a) for a in sys.argv[3:]])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lin-gap":   # usage: lin-gap arch seed [seed ...]   (round 5: the LINEAR-REGIME recipe, workloads/synth.py LIN_GAMMA -> tests/golden/lin_<tag>.npz + ref16_lin_<tag>.npz)
        cond_gap_golden(sys.argv[2], [int(a) for a in sys.argv[3:]], variant="lin", min_dets=16)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "photo":
        if not os.path.exists(os.path.join(HERE, "bus.png")):
            photo_pngs()
        seeds = [int(a) for a in sys.argv[2:] if a.isdigit()]
        for a in [a for a in sys.argv[2:] if not a.isdigit()] or ["yolov5_darknet_pan_s_r60"]:
            photo_golden(a, seeds or range(0, 30))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cond":
        seeds = [int(a) for a in sys.argv[2:] if a.isdigit()]   # usage: cond [arch ...] [seed ...]
        for a in [a for a in sys.argv[2:] if not a.isdigit()] or list(COND_TAGS):
            cond_golden(a, seeds or range(0, 40))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "spread-more":   # usage: spread-more arch seed [seed ...]
        print("accepted seeds:", spread_more(sys.argv[2], [int(a) for a in sys.argv[3:]]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "spread":   # usage: spread [arch ...] [seed ...]
        seeds = [int(a) for a in sys.argv[2:] if a.isdigit()]
        force = "--force" in sys.argv
        for a in [a for a in sys.argv[2:] if not a.isdigit() and a != "--force"] or ["yolov5_darknet_pan_s_r60", "yolov5_darknet_pan_m_r60"]:
            spread_golden(a, seeds or range(0, 40), force=force)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ref16":   # usage: ref16 [kind:tag ...]
        for a in sys.argv[2:] or ["cond:s", "cond:n", "cond:m", "photo:s"]:
            ref16_golden(*a.split(":"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cond-eval":
        cond_evaluate(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    letterbox_golden()
    anchors_decode_golden()
    e2e_golden()
    print("golden written to", HERE)
