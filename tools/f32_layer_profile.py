"""Per-launch table and throughput of the fp32 mode (`YOLOv5.set_compute_dtype(torch.float32)`) on a BASELINE config's workload.

    python tools/f32_layer_profile.py --config c2 [--depth 3] [--steps 12] [--tune] > gpurun_out/f32_layers_c2.csv

Per conv launch (HIP events around each op of the recorded plan, one batch in flight: `ymi_plan_profile`): tile, duration, TFLOP/s and the fraction of
the launch's own bound max(flops / 157.3 TFLOP/s, bytes / 8 TB/s) -- the f32-input MFMA peak and 4-byte elements (MI355X_MICROARCH.md "Matrix cores").
`--tune` times every fp32 tile on every conv of the plan (engine.Plan autotune) and prints the candidates' times: the data the library's shape rule
(csrc/conv_f32_pipe.hip conv_f32_pick_tile) is fitted to.  YOLORT_AMD_F32_V1=1 selects the register-staged kernel of rounds 2-4 (the A/B partner).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

F32_PEAK, HBM_PEAK = 157.3e12, 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--depth", type=int, default=3, help="plan instances (batches in flight) of the throughput loop")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    if a.tune:
        os.environ["YOLORT_AMD_AUTOTUNE"] = "1"
    from yolort_amd import engine
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights

    c = dict(bench.CONFIGS[a.config])
    batch = a.batch or (c["batch"] if c["size"] <= 640 else min(c["batch"], 8))
    dev = torch.device("cuda:0")
    kw = dict(size_divisible=64) if c["arch"].endswith("6_r60") else {}
    m = YOLOv5(arch=c["arch"], size=(c["size"], c["size"]), score_thresh=c["score_thresh"], nms_thresh=0.45, detections_per_img=300, **kw)
    m.load_state_dict(synth_weights(m.state_dict(), c["arch"], seed=0, head_gain=c["head_gain"]))
    m = m.to(dev).eval().set_compute_dtype(torch.float32)
    m.model.pipeline_depth = 1
    imgs = [im.to(dev) for im in synth_images(batch, c["size"], c["size"], seed=1)]
    for _ in range(2):
        m.forward(imgs)
    torch.cuda.synchronize()
    e = next(iter(m.model._entries.values()))
    prof = e.plan.profile(iters=5)
    print("op,name,kind,shape,tile,us,tflops,GBps,bound_us,frac_of_bound")
    tot = {"us": 0.0, "bound": 0.0, "flops": 0.0, "bytes": 0.0, "conv_us": 0.0}
    for i, (name, ms, meta) in enumerate(prof):
        us = ms * 1e3
        fl, by = meta.get("flops", 0.0), meta.get("bytes", 0.0)
        bound = max(fl / F32_PEAK, by / HBM_PEAK) * 1e6
        tot["us"] += us
        if meta.get("kind") == "conv":
            tot["bound"] += bound
            tot["flops"] += fl
            tot["bytes"] += by
            tot["conv_us"] += us
        print(f"{i},{name},{meta.get('kind')},{meta.get('shape', '')},{meta.get('tile', '')},{us:.1f},{fl / max(us, 1e-9) / 1e6:.1f},{by / max(us, 1e-9) / 1e3:.0f},{bound:.1f},{bound / max(us, 1e-9):.3f}")
    print(f"# conv launches: {tot['conv_us']:.0f} us / step of {batch} images; per-layer bound {tot['bound']:.0f} us -> frac {tot['bound'] / max(tot['conv_us'], 1e-9):.3f}; "
          f"{tot['flops'] / max(tot['conv_us'], 1e-9) / 1e6:.1f} TFLOP/s = {tot['flops'] / max(tot['conv_us'], 1e-9) / 1e6 / 157.3:.3f} of the f32 MFMA peak; all ops {tot['us']:.0f} us")
    if a.tune:
        print("# autotune candidates (us):")
        for k, v in engine.Plan._TUNE_TIMES.items():
            best = min(v, key=lambda t: v[t])
            print(f"# {k} :: {json.dumps(v)} best {best}")
    # throughput with `depth` batches in flight (the regime bench.py's value is measured in)
    res = {}
    for depth in sorted({1, a.depth}):
        m.model.pipeline_depth = depth
        pend = []
        for _ in range(depth + 1):
            pend.append(m.forward_async(imgs))
        for p in pend:
            p.result()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = []
        for _ in range(a.steps):
            pend.append(m.forward_async(imgs))
            if len(pend) >= depth:
                pend.pop(0).result()
        for p in pend:
            p.result()
        torch.cuda.synchronize()
        ips = batch * a.steps / (time.perf_counter() - t0)
        res[depth] = round(ips, 1)
        print(f"# fp32 mode throughput, {depth} batch(es) in flight: {ips:.1f} img/s ({1e3 * batch / ips:.3f} ms / step)")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"config": a.config, "batch": batch, "images_per_s": res, "conv_us": tot["conv_us"], "bound_us": tot["bound"], "v1": os.environ.get("YOLORT_AMD_F32_V1", "0")}, f)


if __name__ == "__main__":
    main()
