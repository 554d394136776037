"""Profiling driver (run under rocprofv3): the benchmark workload with ONE batch in flight, so that the kernel dispatches of
a step appear in plan order and every kernel runs alone on the GPU.  Writes the plan's op list (name, shape, tile,
algorithmic flops / bytes) next to the trace; tools/layer_table.py joins the two into the per-layer roofline table.

    rocprofv3 --kernel-trace --stats -d DIR -o r -- python tools/profile_serial.py --config c2 --steps 10 --ops gpurun_out/ops_c2.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from yolort_amd.models import YOLOv5  # noqa: E402
from workloads.synth import synth_images, synth_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--ops", default="")
    ap.add_argument("--score-thresh", type=float, default=None, help="override the config's threshold (e.g. 0.999: a head without candidates)")
    ap.add_argument("--fp32", action="store_true", help="the fp32 mode (fp32 storage, f32-input MFMA: csrc/conv_f32_pipe.hip) instead of the config's 16-bit type")
    a = ap.parse_args()
    c = dict(bench.CONFIGS[a.config])
    if a.score_thresh is not None:
        c["score_thresh"] = a.score_thresh
    dev = torch.device("cuda:0")
    dtype = torch.float16 if c["dtype"] == "fp16" else torch.bfloat16
    kw = dict(size_divisible=64) if c["arch"].endswith("6_r60") else {}
    m = YOLOv5(arch=c["arch"], size=(c["size"], c["size"]), score_thresh=c["score_thresh"], nms_thresh=0.45, detections_per_img=300, **kw)
    m.load_state_dict(synth_weights(m.state_dict(), c["arch"], seed=0, head_gain=c["head_gain"]))
    if a.fp32:
        dtype = torch.float32
        m = m.to(dev).eval().set_compute_dtype(torch.float32)
        m.model.pipeline_depth = 1
        if c["size"] > 640:
            c["batch"] = min(c["batch"], 8)
    else:
        m = m.to(dev).to(dtype).eval()
    m.model.use_graph = False   # per-kernel launches: the same kernels in the same order as the hipGraph replay, each a dispatch of its own for the tracer
    if c["shapes"] == "dynamic":
        imgs = [synth_images(1, *bench.C3_SHAPES[i % 8], seed=1 + i)[0].to(dev).to(dtype) for i in range(c["batch"])]
    else:
        imgs = [im.to(dev).to(dtype) for im in synth_images(c["batch"], c["size"], c["size"], seed=1)]
    for _ in range(3):   # plan build, capacity growth, clocks
        m.forward(imgs)
    torch.cuda.synchronize()
    e = next(iter(m.model._entries.values()))
    if a.ops:
        ops = [{"name": n, **meta} for n, meta in zip(e.plan.names, e.plan.meta)]
        if e.plan.stem_body1_fusable() and (c["shapes"] != "dynamic" or e.plan.fuse_stem):   # ops 0 and 1 run as ONE launch (ymi_stem_body1_planar / ymi_plan_set_fuse_stem): one table row, both layers' algorithmic work
            o0, o1 = ops[0], ops[1]
            ops = [{**o0, "name": o0["name"] + "+" + o1["name"].split(".")[-1], "flops": o0["flops"] + o1["flops"], "bytes": o0["bytes"] + o1["bytes"],
                    "ref_convs": 2, "tile": -2, "shape": "stem 3->32 k6 s2 + 32->64 k3 s2 (one launch)"}] + ops[2:]
        with open(a.ops, "w") as f:
            json.dump({"config": a.config, "steps": a.steps, "batch": c["batch"], "ops": ops}, f, indent=1)
    for _ in range(a.steps):
        m.forward(imgs)
        torch.cuda.synchronize()
    print("profile_serial done", a.config, a.steps, flush=True)


if __name__ == "__main__":
    main()
