"""Experiment: sub-batches on several streams (tuning aid).  usage: python tools/substream_bench.py <sub_batch> <n_streams> <steps>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolort_amd.models import YOLOv5
from workloads.synth import synth_images, synth_weights
sub, ns, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
arch = "yolov5_darknet_pan_s_r60"
m = YOLOv5(arch=arch, score_thresh=0.25)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0))
m = m.to(dev).half().eval()
m.model.pipeline_depth = 2 * ns
imgs = [im.to(dev).half() for im in synth_images(sub, 640, 640, seed=1)]
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
def run(k):
    pend = []
    for i in range(k):
        with torch.cuda.stream(streams[i % ns]):
            pend.append(m.forward_async(imgs))
        if len(pend) > 2 * ns - 1:
            pend.pop(0).result()
    for p in pend: p.result()
run(3 * ns); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"sub_batch={sub} streams={ns}: {steps * sub / dt:.1f} img/s ({dt / steps * 1e3:.3f} ms per sub-batch)")
