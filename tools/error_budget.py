"""TEST / ANALYSIS INFRASTRUCTURE (CPU, oracle only -- nothing here is on the product path).

Per-layer error budget of 16-bit STORAGE on a conditioned golden workload: the oracle is evaluated with the input and the folded weights
of ONE layer at a time rounded to fp16 / bf16 (every other layer exact fp32), then with growing prefixes / suffixes of the layer list, and each evaluation is
compared with the exact fp32 evaluation:

  * relative rms error of the head logits (box rows / objectness row / class rows), i.e. what the layer's rounding alone contributes at the output;
  * the detection-level figures north_star is written in: min IoU and max |dscore| of the paired detections.

The single-layer contributions add in variance (independent roundings); the table shows which layers would have to keep fp32 storage for the box
tolerance 1 - 1e-3 to hold, and `hybrids` evaluates those candidates directly.

usage: python tools/error_budget.py s [fp16|bf16]   -> profiles/r04_error_budget_<tag>_<dtype>.csv / .json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import yolov5_oracle as O  # noqa: E402
from workloads.synth import cond_images, conditioned_weights  # noqa: E402

ARCH = {"n": "yolov5_darknet_pan_n_r60", "s": "yolov5_darknet_pan_s_r60", "m": "yolov5_darknet_pan_m_r60", "l6": "yolov5_darknet_pan_l6_r60"}


def template(arch):
    try:
        from oracle.make_synth_bn import reference_template
        return reference_template(arch)
    except Exception:
        from yolort_amd.models import yolo as Y
        return Y.__dict__[arch]().state_dict()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "s"
    dname = sys.argv[2] if len(sys.argv) > 2 else "fp16"
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[dname]
    arch = ARCH[tag]
    z = np.load(os.path.join(ROOT, "tests", "golden", f"cond_{tag}.npz"))
    meta = json.loads(str(z["meta"]))
    S, thr, seed = meta["S"], meta["thr"], meta["seed"]
    div = 64 if arch.endswith("6_r60") else 32
    sd = conditioned_weights(template(arch), arch, seed)
    sd = {(k if k.startswith("model.") else "model." + k): v for k, v in sd.items()}
    imgs = cond_images(arch, seed)
    torch.set_num_threads(os.cpu_count() or 8)

    def run(select):
        O.EMULATE.dtype, O.EMULATE.select = (dt if select is not None else None), select
        try:
            with torch.no_grad():
                dets, st = O.yolov5_forward(imgs, sd, size=(S, S), size_divisible=div, score_thresh=thr, return_stages=True)
                ho = st["head"]
        finally:
            O.EMULATE.dtype, O.EMULATE.select = None, None
        logits = torch.cat([h.flatten(0, 3) for h in ho]).double()
        return logits, [{k: v.numpy() for k, v in d.items()} for d in dets]

    layers = []
    O.TRACE.hook = lambda p, x, s, pad: layers.append(p)
    try:
        with torch.no_grad():
            O.yolov5_forward(imgs[:1], sd, size=(S, S), size_divisible=div, score_thresh=thr)
    finally:
        O.TRACE.hook = None
    L0, D0 = run(None)
    rms = lambda a: float(np.sqrt((a ** 2).mean()))  # noqa: E731
    scale = {"box": rms((L0[:, :4] - L0[:, :4].mean(0)).numpy()), "obj": rms((L0[:, 4] - L0[:, 4].mean()).numpy()), "cls": rms((L0[:, 5:] - L0[:, 5:].mean(0)).numpy())}

    def measure(select, name):
        L, D = run(select)
        e = (L - L0).numpy()
        c = bench.direct_checks(D0, D, thr, score_eps=0.1, iou_min=0.5)
        row = {"layers_in_16bit": name, "box_logit_rms_err": rms(e[:, :4]), "box_logit_max_err": float(np.abs(e[:, :4]).max()), "obj_logit_rms_err": rms(e[:, 4]),
               "cls_logit_rms_err": rms(e[:, 5:]), "rel_box": rms(e[:, :4]) / scale["box"], "rel_obj": rms(e[:, 4]) / scale["obj"],
               "paired": c["paired"], "ref_dets": c["ref_dets"], "min_iou": c["min_iou"], "iou_deficit": round(1 - c["min_iou"], 6), "max_dscore": c["max_dscore"]}
        print(json.dumps(row), flush=True)
        return row

    rows = [measure(lambda p: True, "ALL")]
    for lay in layers:
        rows.append(measure(lambda p, lay=lay: p == lay, lay))
    n = len(layers)
    for k in sorted(set([n // 8, n // 4, n // 2, 3 * n // 4, n - 3])):   # the first k layers exact (fp32 storage), the rest 16-bit -- and the reverse
        tail = set(layers[k:])
        rows.append(measure(lambda p, t=tail: p in t, f"all but the first {k}"))
        headset = set(layers[:k])
        rows.append(measure(lambda p, t=headset: p in t, f"only the first {k}"))
    # the hybrid VERDICT r3 names: fp32 storage for what feeds the head (the three PAN outputs = the inputs of the head convolutions) and for the last C3s' cv3
    heads = {p for p in layers if ".head." in p}
    cv3 = {p for p in layers if p.endswith(".cv3") and ".layer_blocks." in p}
    rows.append(measure(lambda p: p not in heads, "all but the head convolutions (PAN outputs stored in fp32)"))
    rows.append(measure(lambda p: p not in heads and p not in cv3, "all but the head convolutions and the PAN layer_blocks' cv3"))
    out = os.path.join(ROOT, "profiles", f"r04_error_budget_{tag}_{dname}")
    with open(out + ".json", "w") as f:
        json.dump({"arch": arch, "dtype": dname, "seed": seed, "logit_scale_rms": scale, "layers": layers, "rows": rows}, f, indent=1)
    keys = list(rows[0].keys())
    with open(out + ".csv", "w") as f:
        f.write(f"# per-layer error budget of {dname} storage, conditioned {arch} (seed {seed}), 4 images; logit spread (rms about the channel mean): {scale}\n")
        f.write(",".join(keys) + "\n")
        for r in rows:
            f.write(",".join(str(r[k]) if not isinstance(r[k], float) else f"{r[k]:.4g}" for k in keys) + "\n")
    single = [r for r in rows[1:1 + n]]
    tot = float(np.sqrt(sum(r["box_logit_rms_err"] ** 2 for r in single)))
    print(f"sum in quadrature of the single-layer box-logit errors: {tot:.3e}; all layers together: {rows[0]['box_logit_rms_err']:.3e}")


if __name__ == "__main__":
    main()
