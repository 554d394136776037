"""BUILD CONTAINER ONLY: times the UNMODIFIED reference (`yolort.models.YOLOv5(...).predict(batch)`, /root/reference, fp32 eager CPU) on the bench workloads and
commits the figures to profiles/reference_cpu_baseline.json (VERDICT r3 item 2 / SURVEY.md 8d).  bench.py quotes the record next to its live CPU leg on boxes
where /root/reference does not exist (the GPU box), labelled with this host's core count.

usage: python tools/reference_cpu_baseline.py [c1 c2 c3 c5]
"""
import json
import os
import platform
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from workloads.synth import synth_images, synth_weights  # noqa: E402


def main():
    cfgs = sys.argv[1:] or ["c2"]
    path = os.path.join(ROOT, "profiles", "reference_cpu_baseline.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for cfg in cfgs:
        c = bench.CONFIGS[cfg]
        args = SimpleNamespace(arch=c["arch"], size=c["size"], score_thresh=c["score_thresh"], cpu_threads=os.cpu_count(), config=cfg)
        from yolort_amd.models import yolo as Y
        kw = dict(size_divisible=64) if c["arch"].endswith("6_r60") else {}
        from yolort_amd.models import YOLOv5
        tmpl = YOLOv5(arch=c["arch"], size=(c["size"], c["size"]), **kw).state_dict()
        sd = synth_weights(tmpl, c["arch"], seed=0, head_gain=c["head_gain"])
        if c["shapes"] == "dynamic":
            imgs = [synth_images(1, *bench.C3_SHAPES[i % len(bench.C3_SHAPES)], seed=1 + i)[0] for i in range(min(c["batch"], 8))]
        else:
            imgs = list(synth_images(min(c["batch"], 32), c["size"], c["size"], seed=1))
        rec = bench.reference_cpu_baseline(args, sd, imgs, budget_s=25.0, cores=os.cpu_count())
        rec["host"] = f"build container, {os.cpu_count()} cpus ({platform.processor() or platform.machine()}), torch {torch.__version__}"
        rec["workload"] = f"{cfg}: {c['arch']} {c['size']}^2, score_thresh {c['score_thresh']}, synthetic weights (head_gain {c['head_gain']}), " + ("the 8 cycled image sizes" if c["shapes"] == "dynamic" else "fixed-size images")
        out[cfg] = rec
        print(cfg, json.dumps(rec), flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
