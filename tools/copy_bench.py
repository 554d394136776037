"""HBM streaming rate of plain 16-byte copies as a function of the access shape (tuning aid): what the concat-eliminating channel
slices cost.  A chain of copies over DISTINCT large buffers (nothing stays in the 256 MB MALL between launches):
  contiguous      (N, H, W, C) -> (N, H, W, C)
  slice -> slice  channels [0, C) of a 2C-wide buffer -> channels [C, 2C) of another 2C-wide buffer (64-byte halves of 128-byte rows at C = 32)
usage: python tools/copy_bench.py [C] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolort_amd import engine

C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
n, h, w = 32, 160, 160
NB = 8   # buffers in the chain: 8 x 52 MB (C = 32) / 105 MB (2C) -- more than the MALL holds
for mode in ("contiguous", "slice->slice", "contiguous->slice", "slice->contiguous"):
    plan = engine.Plan(dev, torch.float16)
    src_wide = mode.startswith("slice")
    dst_wide = mode.endswith("slice")
    bufs_s = [plan.alloc(n, h, w, 2 * C if src_wide else C) for _ in range(NB)]
    bufs_d = [plan.alloc(n, h, w, 2 * C if dst_wide else C) for _ in range(NB)]
    for b in bufs_s:
        b.base.normal_()
    for i in range(NB):
        x = bufs_s[i].slice_c(0, C) if src_wide else bufs_s[i]
        y = bufs_d[i].slice_c(C, C) if dst_wide else bufs_d[i]
        plan.copy(x, y)
    plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * NB)
    nbytes = 2 * n * h * w * C * 2
    print(f"copy C={C} {mode:22s}: {ms * 1e3:7.1f} us per launch, {nbytes / 1e6:.0f} MB moved -> {nbytes / ms / 1e9:.2f} TB/s")
