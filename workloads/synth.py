"""Seeded synthetic "trained-like" weights (data generation only; no compute path).

There are no pretrained checkpoints offline and the reference's default init makes every score
vanish, so benchmarks and parity tests use a seeded recipe:

  * conv weights        ~ N(0, 1/fan_in), rounded to fp16-representable values
  * BatchNorm affine    weight = 0.75 + 0.5 U, bias = 0.2 N
  * BatchNorm statistics loaded from a committed calibration file
    ``workloads/data/synth_bn_<arch>_s<seed>.npz`` (produced once, on CPU, by
    ``oracle/make_synth_bn.py`` -- batch statistics of a seeded calibration batch)
  * head convs          weight ~ N(0, (g/sqrt(Cin))^2) for the obj/cls outputs and a quarter of that for
                        the 4 box outputs (keeps boxes anchor-sized instead of degenerate),
                        bias box 0 / obj `obj_bias` / cls `cls_bias`

The same state_dict is loaded into the reference (oracle) and into this package, exactly like a
real yolort checkpoint would be (reference loader: yolort/models/yolo.py:259-263).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _fp16_round(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).astype(np.float32)


def synth_state_dict(
    template: Dict[str, torch.Tensor],
    seed: int = 0,
    head_gain: float = 0.5,
    obj_bias: float = -4.0,
    cls_bias: float = -1.0,
    num_outputs: int = 85,
    bn_gamma=(0.75, 1.25),
    obj_gain: float = 1.0,
    cls_prior=None,
) -> Dict[str, torch.Tensor]:
    """Fill a state_dict (keys/shapes taken from `template`) with the seeded recipe.

    BatchNorm running statistics are left at (0, 1); see `load_synth_bn`.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, torch.Tensor] = {}
    for key, t in template.items():
        shape = tuple(t.shape)
        if key.endswith("num_batches_tracked"):
            v = np.zeros(shape, np.int64)
            out[key] = torch.from_numpy(v)
            continue
        if key.endswith(".conv.weight") or (len(shape) == 4 and key.endswith(".weight") and ".head." not in key):   # (the second form: BottleneckCSP's bare cv2 / cv3 of the r3.1 models)
            fan_in = shape[1] * shape[2] * shape[3]
            v = _fp16_round(rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in))
        elif key.endswith(".bn.weight"):
            v = (np.float32(bn_gamma[0]) + np.float32(bn_gamma[1] - bn_gamma[0]) * rng.random(shape, dtype=np.float32)).astype(np.float32)
        elif key.endswith(".bn.bias"):
            v = (0.2 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif key.endswith(".bn.running_mean"):
            v = np.zeros(shape, np.float32)
        elif key.endswith(".bn.running_var"):
            v = np.ones(shape, np.float32)
        elif ".head." in key and key.endswith(".weight"):
            v = rng.standard_normal(shape, dtype=np.float32) * (head_gain / np.sqrt(shape[1]))
            if shape[0] % num_outputs:   # a head with another class count: three anchors per level
                num_outputs = shape[0] // 3
            v = v.reshape(shape[0] // num_outputs, num_outputs, *shape[1:])
            v[:, :4] *= 0.25  # box regressors: small logits -> sigmoid near 0.5 -> anchor-sized boxes
            v[:, 4] *= obj_gain  # (1.0 except in the "spread" recipe below)
            v = _fp16_round(v.reshape(shape))
        elif ".head." in key and key.endswith(".bias"):
            if shape[0] % num_outputs:
                num_outputs = shape[0] // 3
            b = np.zeros((shape[0] // num_outputs, num_outputs), np.float32)
            b[:, 4] = obj_bias
            b[:, 5:] = cls_bias if cls_prior is None else np.asarray(cls_prior, np.float32)[None, : num_outputs - 5]
            v = b.reshape(shape)
        else:
            raise KeyError(f"synth recipe does not know key {key}")
        out[key] = torch.from_numpy(np.ascontiguousarray(v))
    return out


def synth_bn_path(arch: str, seed: int) -> str:
    return os.path.join(_DATA, f"synth_bn_{arch}_s{seed}.npz")


def load_synth_bn(sd: Dict[str, torch.Tensor], arch: str, seed: int = 0, path: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """Overwrite the BatchNorm running statistics of `sd` with the committed calibration."""
    path = path or synth_bn_path(arch, seed)
    if not os.path.exists(path):
        raise FileNotFoundError(f"no committed BN calibration for arch={arch} seed={seed}: {path} (run oracle/make_synth_bn.py)")
    z = np.load(path)
    prefix = "model." if any(k.startswith("model.") for k in sd) else ""
    for k in z.files:
        kk = prefix + k
        if kk not in sd:
            raise KeyError(f"calibration key {kk} not in state_dict")
        sd[kk] = torch.from_numpy(z[k].astype(np.float32))
    return sd


def synth_weights(template: Dict[str, torch.Tensor], arch: str, seed: int = 0, **kw) -> Dict[str, torch.Tensor]:
    return load_synth_bn(synth_state_dict(template, seed=seed, **kw), arch, seed)


# ---- the CONDITIONED recipe (round 3): the parity workload that can carry a tolerance --------------------------------------
# With BatchNorm weights around 1 every Conv-BN-SiLU of a random network amplifies a relative perturbation (a rounding) by
# sqrt(chi), chi = gamma^2 E[silu'(z)^2] / Var(silu(z)) = 1.21 for z ~ N(0,1): x10 over the ~25 layers in sequence between the
# image and a head (measured: fp16 storage reaches the logits as 4.4e-3 relative, x2500 for an fp32 rounding -- DESIGN.md section 2).
# chi >= 1 for any variance-normalising network (Gaussian Poincare inequality) and -> 1 as the activation becomes linear:
# gamma in [0.3, 0.6] gives chi = 1.1 (measured: 1.5e-3, the no-amplification floor of 25 fp16 roundings).  Head gain 1.0 and an
# objectness bias TUNED on a seeded batch so that a few dozen (anchor, class) pairs per image pass the threshold -- sparse,
# well separated detections like a trained detector's, instead of 1000 near-tied candidates per image competing in the NMS.
COND_GAMMA = (0.3, 0.6)
COND_HEAD_GAIN = 1.0


COND_SIZE = {"yolov5_darknet_pan_n_r60": 640, "yolov5_darknet_pan_s_r60": 640, "yolov5_darknet_pan_m_r60": 1280, "yolov5_darknet_pan_l6_r60": 1280,
             "yolov5_darknet_pan_s_r40": 640, "yolov5_darknet_pan_s_r31": 640}   # (round 5: the legacy releases, Focus stem; r3.1: BottleneckCSP / Hardswish / LeakyReLU)


def cond_images(arch: str, seed: int = 0):
    """the seeded tuning / parity batch of the conditioned workload: four U[0,1) images of mixed shapes (the letterbox is exercised too)"""
    S = COND_SIZE[arch]
    shapes = [(S, S), (S * 3 // 4, S), (S, S * 2 // 3 + 1), (S * 5 // 4 + 3, S * 3 // 2 + 10)]   # identity, two paddings, one down-scale
    return [synth_images(1, h, w, seed=5000 + 10 * seed + i)[0] for i, (h, w) in enumerate(shapes)]


# ---- the SPREAD recipe (round 4): the conditioned network with a detection head whose scores cover 0.25 ... 0.9 -----------------------------
# The conditioned recipe thresholds the far tail of a Gaussian logit: every detection sits within a few hundredths of the threshold, so a 16-bit
# comparison with a score tolerance of that size can excuse all of them.  Here the objectness row has gain SPREAD_OBJ_GAIN (its
# sigmoid reaches 0.9 at the hottest pixels), SPREAD_HOT classes carry a prior (class bias) spread over sigmoid 0.62 ... 0.95 and the others are
# off (bias -6): a firing anchor yields a handful of labels whose scores spread with the class prior and the objectness.  The score threshold of a
# golden is then placed in a GAP of the reference's score list (tests/golden/make_golden.py spread): no detection is near the cut, so a 16-bit
# evaluation has to reproduce every one of them -- nothing is excused.
SPREAD_OBJ_GAIN = 4.0
SPREAD_HOT = 8
SPREAD_TARGET = 6      # anchors per image with sigmoid(objectness) > SPREAD_OBJ_LEVEL on the tuning batch
SPREAD_OBJ_LEVEL = 0.4


def spread_cls_prior(seed: int, num_classes: int = 80) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(77000 + seed))
    prior = np.full(num_classes, -6.0, np.float32)
    prior[rng.permutation(num_classes)[:SPREAD_HOT]] = np.linspace(0.5, 3.0, SPREAD_HOT).astype(np.float32)
    return prior


def spread_images(arch: str, seed: int = 0):
    """the seeded batch of the spread workload: four U[0,1) images -- identity, two paddings (no resampling), one down-scale"""
    S = COND_SIZE[arch]
    shapes = [(S, S), (S * 3 // 4, S), (S, S * 7 // 8), (S * 5 // 4 + 3, S * 3 // 2 + 10)]
    return [synth_images(1, h, w, seed=9000 + 10 * seed + i)[0] for i, (h, w) in enumerate(shapes)]


# ---- the LINEAR-REGIME recipe (round 5): the conditioned network with BatchNorm weights in [0.08, 0.16] ------------------------------------------------
# chi -> 1 as the pre-activation spread shrinks (silu is locally affine around each channel's BatchNorm bias): with gamma ~ 0.1 a rounding is no longer amplified
# on its way through the 80 (yolov5m) / 135 (yolov5l6) layers, so what a 16-bit evaluation of the DEEP networks loses is the plain sum of its roundings.  This is the
# workload on which the 16-bit production path of BASELINE configs[2] / [4] is held to an ABSOLUTE tolerance (tests/test_golden_gpu.py): with the
# conditioned recipe's gamma in [0.3, 0.6] the reference's OWN .half() run of yolov5l6 pairs 6 of 27 detections with its fp32 run.
LIN_GAMMA = (0.08, 0.16)


def cond_bn_path(arch: str, seed: int, variant: str = "cond") -> str:
    return os.path.join(_DATA, f"synth_bn_{arch}_s{seed}_{variant}.npz")


def conditioned_weights(template: Dict[str, torch.Tensor], arch: str, seed: int = 0, path: Optional[str] = None, variant: str = "cond") -> Dict[str, torch.Tensor]:
    """The conditioned recipe: `synth_state_dict` with COND_GAMMA / COND_HEAD_GAIN, the BatchNorm statistics and the tuned objectness
    bias from the committed calibration file (oracle/make_synth_bn.py --cond).  variant "photo": the same weights calibrated and tuned on the
    reference's two asset photos (tests/golden/{bus,zidane}.png) instead of the seeded noise batch."""
    path = path or cond_bn_path(arch, seed, variant)
    if not os.path.exists(path):
        raise FileNotFoundError(f"no committed conditioned calibration for arch={arch} seed={seed}: {path} (run oracle/make_synth_bn.py --cond)")
    z = np.load(path)
    extra = dict(obj_gain=SPREAD_OBJ_GAIN, cls_prior=spread_cls_prior(seed)) if variant == "spread" else {}
    head_gain = float(z["__head_gain__"]) if "__head_gain__" in z.files else COND_HEAD_GAIN   # (linear-regime recipe: the head is scaled so that its logits spread like the conditioned recipe's)
    sd = synth_state_dict(template, seed=seed, head_gain=head_gain, obj_bias=float(z["__obj_bias__"]), bn_gamma=LIN_GAMMA if variant == "lin" else COND_GAMMA, **extra)
    prefix = "model." if any(k.startswith("model.") for k in sd) else ""
    for k in z.files:
        if k.startswith("__"):
            continue
        if prefix + k not in sd:
            raise KeyError(f"calibration key {prefix + k} not in state_dict")
        sd[prefix + k] = torch.from_numpy(z[k].astype(np.float32))
    return sd


def synth_images(n: int, h: int = 640, w: int = 640, seed: int = 1) -> torch.Tensor:
    """Seeded U[0,1) images (N,3,H,W) fp32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random((n, 3, h, w), dtype=np.float32))
