"""Letterbox alone on the GPU (tuning aid): the C3 batch (64 images, the 8 cycled shapes of SURVEY 8d, bf16 -> 1280x1280 canvas)
or the C2-like dynamic batch, HIP events around `reps` launches with nothing else on the device.
usage: python tools/letterbox_bench.py [c3|c2dyn] [reps]     (YOLORT_AMD_LETTERBOX selects the kernel: pixel | 1 | 2 | 4)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolort_amd.engine import View
from yolort_amd.models.transform import YOLOTransform
from workloads.synth import synth_images

C3_SHAPES = [(1080, 1920), (720, 1280), (1920, 1080), (1080, 810), (960, 1280), (1281, 1279), (641, 480), (375, 500)]
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
if which == "c3":
    n, S, dt = 64, 1280, torch.bfloat16
else:
    n, S, dt = 32, 640, torch.float16
tr = YOLOTransform(S, S)
imgs = [synth_images(1, *C3_SHAPES[i % 8], seed=100 + i)[0].to(dev).to(dt) for i in range(n)]
(hb, wb), sizes, pads = tr.geometry([tr.image_hw(im) for im in imgs])
t = torch.empty(n * hb * wb * 4, device=dev, dtype=dt)
v = View(t, 0, n, hb, wb, 4, 4)
for _ in range(3):
    tr.letterbox_into(imgs, v, sizes, pads)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    tr.letterbox_into(imgs, v, sizes, pads)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
esz = 2
nbytes = sum(3 * im.shape[1] * im.shape[2] * esz for im in imgs) + n * hb * wb * 4 * esz
if os.environ.get("LB_COPY_REF"):
    t2 = torch.empty_like(t)
    for _ in range(3):
        t2.copy_(t)
    e0.record()
    for _ in range(reps):
        t2.copy_(t)
    e1.record()
    torch.cuda.synchronize()
    cms = e0.elapsed_time(e1) / reps
    print(f"reference: device copy of the canvas buffer ({t.numel() * esz / 1e6:.0f} MB read + written): {2 * t.numel() * esz / cms / 1e9:.3f} TB/s")
dbg = os.environ.get("YOLORT_AMD_LB_DEBUG", "0")
print(f"letterbox {which} debug={dbg} kernel={os.environ.get('YOLORT_AMD_LETTERBOX', 'default')}: {ms * 1e3:.1f} us per batch, {nbytes / 1e6:.1f} MB algorithmic -> "
      f"{nbytes / ms / 1e9:.3f} TB/s = {nbytes / ms / 1e9 / 8.0:.3f} of 8 TB/s; checksum {float(v.as_tensor().float().sum()):.6e}")
