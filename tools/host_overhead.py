"""Tuning aid: host-side cost of one forward_async() enqueue and one result() collection (yolov5s bs 32, planar stem path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolort_amd.models import YOLOv5
from workloads.synth import synth_images, synth_weights

dev = torch.device("cuda:0")
arch = "yolov5_darknet_pan_s_r60"
m = YOLOv5(arch=arch, size=(640, 640), score_thresh=0.25)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0))
m = m.to(dev).half().eval()
m.model.use_graph = os.environ.get("YOLORT_AMD_GRAPH", "0") == "1"
imgs = [im.to(dev).half() for im in synth_images(32, 640, 640, seed=1)]
for _ in range(6):
    m.forward_async(imgs).result()
torch.cuda.synchronize()
t0 = time.perf_counter()
pend = [m.forward_async(imgs) for _ in range(3)]
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
outs = [p.result() for p in pend]
t3 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 3:.3f} ms per batch; collect (GPU already idle) {1e3 * (t3 - t2) / 3:.3f} ms per batch")
