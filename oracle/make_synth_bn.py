"""TEST INFRASTRUCTURE -- produces workloads/data/synth_bn_<arch>_s<seed>.npz.

Runs the oracle's conv stack once in calibration mode (every Conv-BN takes the batch statistics of
its own conv output on a seeded calibration batch) and stores only the
BatchNorm running statistics (a few hundred KB).  Key/shape templates come from the reference
itself when /root/reference is importable (this container), else from yolort_amd.

usage: python oracle/make_synth_bn.py [arch ...]   (default: n s m l6 r60 variants)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import yolov5_oracle as O  # noqa: E402
from workloads.synth import (COND_GAMMA, LIN_GAMMA, COND_HEAD_GAIN, COND_SIZE, SPREAD_OBJ_GAIN, SPREAD_OBJ_LEVEL, SPREAD_TARGET, cond_bn_path, cond_images, spread_cls_prior,  # noqa: E402
                                    spread_images, synth_bn_path, synth_images, synth_state_dict)


def reference_template(arch: str):
    from oracle.reference_loader import load_reference
    yolo = load_reference().models.yolo
    return yolo.__dict__[arch]().state_dict()


# calibration batch per architecture: (resolution, images).  The statistics of the deep, low-resolution layers are heavy-tailed
# and border-dominated, so a network is calibrated AT the resolution it is benchmarked at, on enough images for stable
# per-channel variances (round 1 calibrated everything on 2 images of 320x320: yolov5m / yolov5l6 then ran 10-1000x hot at
# 1280x1280 -- activations up to 3e4, ~10^6 candidates per image; n / s at 640x640 were and stay fine).
CALIB = {"yolov5_darknet_pan_m_r60": (1280, 4), "yolov5_darknet_pan_l6_r60": (1280, 4)}


def calibrate(arch: str, seed: int = 0, calib_hw: int = 0, calib_n: int = 0):
    if not calib_hw:
        calib_hw, calib_n = CALIB.get(arch, (320, 2))
    try:
        tmpl = reference_template(arch)
    except Exception:   # /root/reference not importable (GPU box): same keys / shapes from this package
        from yolort_amd.models import yolo as Y
        tmpl = Y.__dict__[arch]().state_dict()
    sd = synth_state_dict(tmpl, seed=seed)
    x = synth_images(calib_n, calib_hw, calib_hw, seed=1000 + seed)
    O.CALIB.active = True
    try:
        with torch.no_grad():
            feats = O.backbone(x, sd, "backbone")
    finally:
        O.CALIB.active = False
    stats = {k: v.numpy().astype(np.float32) for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    os.makedirs(os.path.dirname(synth_bn_path(arch, seed)), exist_ok=True)
    np.savez(synth_bn_path(arch, seed), **stats)
    print(arch, "features std", [round(float(f.std()), 3) for f in feats], "absmax", [round(float(f.abs().max()), 1) for f in feats],
          "->", synth_bn_path(arch, seed), os.path.getsize(synth_bn_path(arch, seed)) // 1024, "KB")


# ---- the conditioned recipe (workloads/synth.py: COND_*): BatchNorm statistics at the resolution the parity workload runs at, and the
# objectness bias tuned so that about COND_TARGET (anchor, class) pairs per image pass the threshold on the seeded tuning batch ----
COND = {a: (s, 2) for a, s in COND_SIZE.items()}   # (resolution, calibration images)
COND_TARGET = 10
COND_THRESH = 0.25


def photo_images():
    """the reference's two asset photos as decoded (tests/golden/*.png, written by tests/golden/make_golden.py from test/assets/*.jpg), CHW fp32 / 255"""
    from PIL import Image
    out = []
    for name in ("bus", "zidane"):
        a = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", name + ".png")).convert("RGB"))
        out.append(torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32) / 255.0)
    return out


def calibrate_conditioned(arch: str, seed: int = 0, target: int = COND_TARGET, thr: float = COND_THRESH, photo: bool = False, spread: bool = False, lin: bool = False):
    S, n = COND[arch]
    try:
        tmpl = reference_template(arch)
    except Exception:
        from yolort_amd.models import yolo as Y
        tmpl = Y.__dict__[arch]().state_dict()
    extra = dict(obj_gain=SPREAD_OBJ_GAIN, cls_prior=spread_cls_prior(seed)) if spread else {}
    sd = synth_state_dict(tmpl, seed=seed, head_gain=COND_HEAD_GAIN, obj_bias=0.0, bn_gamma=LIN_GAMMA if lin else COND_GAMMA, **extra)
    div = 64 if arch.endswith("6_r60") else 32
    imgs = photo_images() if photo else (spread_images(arch, seed) if spread else cond_images(arch, seed))
    batch, _ = O.letterbox(imgs, S, S, div)
    # yolov5l6 (round 3): with BatchNorm statistics taken on full-frame noise, the letterboxed, resized evaluation images drive the deep P6 network 10 - 200 x out of
    # its calibrated range (PAN output rms 5 / 24 / 91 / 227 per level, head logits of +-900: every image empty or saturated, tests/golden/cond_l6_search.txt), so
    # its statistics come from the evaluation batch itself, like the photo variant's; the committed n / s / m calibrations are untouched
    calib = batch if (photo or spread or lin or arch.endswith("6_r60")) else synth_images(n, S, S, seed=1000 + seed)
    O.CALIB.active = True
    try:
        with torch.no_grad():
            O.backbone(calib, sd, "backbone")
    finally:
        O.CALIB.active = False
    with torch.no_grad():
        feats = O.backbone(batch, sd, "backbone")
        ho = O.head(feats, sd, "head")
    head_gain = COND_HEAD_GAIN
    if lin:   # the linear-regime features are small (BatchNorm weights ~0.1): scale the head so that the objectness logit spreads like the conditioned recipe's (std 1)
        head_gain = round(float(COND_HEAD_GAIN / torch.cat([h[..., 4].flatten() for h in ho]).std()), 3)
        head = synth_state_dict(tmpl, seed=seed, head_gain=head_gain, obj_bias=0.0, bn_gamma=LIN_GAMMA)
        for k in head:
            if ".head." in k:
                sd[k] = head[k]
        with torch.no_grad():
            ho = O.head(feats, sd, "head")
    obj = torch.cat([h[..., 4].flatten() for h in ho])
    pc = torch.sigmoid(torch.cat([h[..., 5:].flatten(0, -2) for h in ho]))
    lo, hi = -14.0, 6.0
    lo, hi = (-40.0, 6.0) if spread else (lo, hi)
    for _ in range(48):   # bisection on the candidate count (monotone in the bias)
        mid = 0.5 * (lo + hi)
        if spread:   # the spread recipe counts hot ANCHORS (objectness alone): each yields several labels through the class priors
            count = int((torch.sigmoid(obj + mid) > SPREAD_OBJ_LEVEL).sum())
            over = count > SPREAD_TARGET * len(imgs)
        else:
            over = int((torch.sigmoid(obj + mid)[:, None] * pc > thr).sum()) > target * len(imgs)
        if over:
            hi = mid
        else:
            lo = mid
    bias = round(0.5 * (lo + hi), 3)
    stats = {k: v.numpy().astype(np.float32) for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    stats["__obj_bias__"] = np.float32(bias)
    if lin:
        stats["__head_gain__"] = np.float32(head_gain)
    out = cond_bn_path(arch, seed, "photo" if photo else ("spread" if spread else ("lin" if lin else "cond")))
    np.savez(out, **stats)
    cand = int((torch.sigmoid(obj + bias)[:, None] * pc > thr).sum())
    print(arch, "conditioned" + (" (photos)" if photo else "") + ": obj bias", bias, "candidates", cand, "on", len(imgs), "images ->", out, os.path.getsize(out) // 1024, "KB")


if __name__ == "__main__":
    if "--cond" in sys.argv:
        photo = "--photo" in sys.argv
        spread = "--spread" in sys.argv
        lin = "--lin" in sys.argv
        args = [a for a in sys.argv[1:] if a not in ("--cond", "--photo", "--spread", "--lin")]
        seed = 0
        for a in list(args):
            if a.startswith("--seed="):
                seed = int(a.split("=")[1])
                args.remove(a)
        for a in args or list(COND):
            calibrate_conditioned(a, seed, photo=photo, spread=spread, lin=lin)
        sys.exit(0)
    archs = sys.argv[1:] or ["yolov5_darknet_pan_n_r60", "yolov5_darknet_pan_s_r60", "yolov5_darknet_pan_m_r60", "yolov5_darknet_pan_l6_r60"]
    for a in archs:
        calibrate(a)
