"""GPU micro-benchmark of single convolutions through the C ABI (tuning aid, not part of the product).
usage: python tools/conv_bench.py [case ...]   each case: n,cin,cout,h,w,k,s,p[,tile]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolort_amd import engine

dev = torch.device("cuda:0")
DEFAULT = [
    # yolov5s bs32 layers
    "32,8,32,640,320,stem", "32,32,64,320,320,3,2,1", "32,64,64,160,160,1,1,0", "32,32,32,160,160,3,1,1", "32,64,128,160,160,3,2,1",
    "32,64,64,80,80,3,1,1", "32,128,256,80,80,3,2,1", "32,128,128,40,40,3,1,1", "32,256,512,40,40,3,2,1", "32,256,256,20,20,3,1,1",
    "32,512,256,20,20,1,1,0", "32,1024,512,20,20,1,1,0",
]

def run(case, tiles, iters=20):
    f = case.split(",")
    n, cin, cout, h, w = map(int, f[:5])
    g = torch.Generator().manual_seed(0)
    res = []
    for tile in tiles:
        plan = engine.Plan(dev, torch.float16)
        if f[5] == "stem":
            x = plan.alloc(n, h, w * 2, 4); x.base.normal_()
            wt = torch.randn(cout, 3, 6, 6, generator=g) / 10
            pc = engine.PackedConv(wt, None, None, torch.float16, dev, stem_superpixel=True)
            try:
                plan.conv(x, pc, 2, 2, tile=tile)
            except Exception as e:
                res.append((tile, None, str(e)[:60])); continue
        else:
            k, s, p = map(int, f[5:8])
            x = plan.alloc(n, h, w, cin); x.base.normal_()
            wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
            pc = engine.PackedConv(wt, None, None, torch.float16, dev)
            short = None
            if os.environ.get("RES", "0") == "1" and s == 1:   # a shortcut operand (Bottleneck.add)
                short = plan.alloc(n, h, w, cout); short.base.normal_()
            try:
                plan.conv(x, pc, s, p, tile=tile, act=int(os.environ.get('ACT', '1')), res=short)
            except Exception as e:
                res.append((tile, None, str(e)[:60])); continue
        try:
            plan.run(); torch.cuda.synchronize()
            prof = plan.profile(iters)
        except Exception as e:
            res.append((tile, None, str(e)[:80])); continue
        ms = prof[0][1]; meta = prof[0][2]
        res.append((tile, ms, f"{meta['flops']/ms/1e9:7.1f} TF {meta['bytes']/ms/1e6:7.1f} GB/s"))
    return res

if __name__ == "__main__":
    cases = [a for a in sys.argv[1:] if "," in a] or DEFAULT
    tiles = [int(t) for t in os.environ.get("TILES", "0,11,12,13,14,15,-100").split(",")]
    for c in cases:
        out = run(c, tiles)
        print(f"{c:32s} " + " | ".join(f"t{t}: {'%.3f' % ms if ms is not None else 'ERR'} {info}" for t, ms, info in out), flush=True)
