"""Tuning aid: per-tile s_memtime timeline of the resident-weights 3x3 kernel (conv3x3_res.hip), wave 0 of every block.
build:  bash tools/build_r3_stamps.sh          (instrumented copy of conv3x3_res only, never the shipped library)
run:    YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_r3stamps.so python tools/stamp_r3.py n,cin,cout,h,w [...]
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolort_amd import engine, _lib

dev = torch.device("cuda:0")
lib = _lib.load()
lib.ymi_debug_stamps_r3.restype = C.c_int
lib.ymi_debug_stamps_r3.argtypes = [C.c_void_p, C.c_int]
NAMES = ["wait vmcnt(0) [patch + stores]", "barrier", "DMA issue of the next patch", "shortcut load + acc init", "MFMA loop (+ previous tile's epilogue)", "hand-over"]

for case in sys.argv[1:]:
    n, cin, cout, h, w = map(int, case.split(","))
    plan = engine.Plan(dev, torch.float16)
    x = plan.alloc(n, h, w, cin); x.base.normal_()
    wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    pc = engine.PackedConv(wt, None, None, torch.float16, dev)
    res = None
    if os.environ.get("RES", "0") == "1":
        res = plan.alloc(n, h, w, cout); res.base.normal_()
    plan.conv(x, pc, 1, 1, tile=132, res=res)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    ms = plan.profile(10)[0][1]
    plan.run(); torch.cuda.synchronize()
    st = np.zeros(256 * 128, dtype=np.uint64)
    assert lib.ymi_debug_stamps_r3(st.ctypes.data, st.size) == 0
    st = st.reshape(256, 16, 8).astype(np.int64)
    ntiles = n * ((h + 15) // 16) * ((w + 15) // 16)
    per_block = min(16, ntiles // 256)
    print(f"== {case}: {ms * 1e3:.1f} us, {ntiles} tiles on 256 blocks; cycles (s_memtime) per phase, mean over blocks, per tile ordinal")
    for k in range(per_block):
        d = [(st[:, k, j + 1] - st[:, k, j]).mean() for j in range(6)]
        nxt = (st[:, k + 1, 0] - st[:, k, 6]).mean() if k + 1 < per_block else 0.0
        print(f"   tile {k:2d}: " + "  ".join(f"{v:6.0f}" for v in d) + f"   | total {sum(d):6.0f}  (to next tile top {nxt:4.0f})")
    mid = st[:, 2:per_block - 1]
    d = [(mid[:, :, j + 1] - mid[:, :, j]).mean() for j in range(6)]
    print("   steady state: " + "; ".join(f"{nm} {v:.0f}" for nm, v in zip(NAMES, d)) + f"; tile total {(mid[:, :, 6] - mid[:, :, 0]).mean():.0f}")
