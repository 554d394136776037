"""One 1x1 conv over a CHAIN of distinct buffers (tuning aid): the same launch as tools/conv_bench.py but on NB different
(input, output) pairs in turn, so nothing is MALL-resident between launches -- the condition a layer meets inside the plan.
usage: python tools/stream_chain_bench.py cin cout h w tile[,tile...] [NB]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolort_amd import engine

cin, cout, h, w = map(int, sys.argv[1:5])
tiles = [int(t) for t in sys.argv[5].split(",")]
NB = int(sys.argv[6]) if len(sys.argv) > 6 else 8
dev = torch.device("cuda:0")
n = 32
g = torch.Generator().manual_seed(0)
wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
for tile in tiles:
    for nb in (1, NB):
        plan = engine.Plan(dev, torch.float16)
        pc = engine.PackedConv(wt, None, None, torch.float16, dev)
        xs = [plan.alloc(n, h, w, cin) for _ in range(nb)]
        for x in xs:
            x.base.normal_()
        for x in xs:
            plan.conv(x, pc, 1, 0, tile=tile)
        plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40 // nb
        e0.record()
        for _ in range(reps):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * nb)
        nbytes = n * h * w * (cin + cout) * 2
        print(f"conv {cin}->{cout} @{h}x{w} tile {tile}, {nb} buffer pair(s): {ms * 1e3:6.1f} us per launch, {nbytes / 1e6:.0f} MB -> {nbytes / ms / 1e9:.2f} TB/s")
