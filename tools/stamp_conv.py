"""Tuning aid: per-block s_memtime timeline of the software-pipelined implicit-GEMM kernel.
build the instrumented library first (never the shipped one):
  YOLORT_AMD_BUILD_OUT=$PWD/tools/_bin/libyolort_amd_stamps.so YOLORT_AMD_BUILD_FLAGS=-DYMI_STAMPS python -m yolort_amd._build
run:  YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_stamps.so python tools/stamp_conv.py n,cin,cout,h,w,k,s,p,tile [...]
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolort_amd import engine, _lib

dev = torch.device("cuda:0")
lib = _lib.load()
lib.ymi_debug_stamps.restype = C.c_int
lib.ymi_debug_stamps.argtypes = [C.c_void_p, C.c_int]

for case in sys.argv[1:]:
    plan = engine.Plan(dev, torch.float16)
    if case.startswith("head,"):   # head,n,cin,hw,thr : fused head + decode of one level (80 classes)
        _, n, cin, h, thr = case.split(",")
        n, cin, h, w, k = int(n), int(cin), int(h), int(h), 1
        x = plan.alloc(n, h, w, cin); x.base.normal_()
        wt = torch.randn(288, cin, 1, 1) / cin ** 0.5
        pc = engine.PackedConv(wt, torch.full((288,), -2.0), None, torch.float16, dev)
        pb, pd = plan.post_desc([(h, w)], n, [8.0], [[10, 13, 16, 30, 33, 23]], 80, float(thr), 0.45, 300, 16384 * n)
        plan.head_decode(x, pc, pd, 0)
        cin = cin
    else:
        n, cin, cout, h, w, k, s, p, tile = map(int, case.split(","))
        x = plan.alloc(n, h, w, cin); x.base.normal_()
        wt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        pc = engine.PackedConv(wt, None, None, torch.float16, dev)
        plan.conv(x, pc, s, p, tile=tile)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    ms = plan.profile(10)[0][1]
    plan.run(); torch.cuda.synchronize()
    st = np.zeros(2048 * 128, dtype=np.uint64)
    assert lib.ymi_debug_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(2048, 128).astype(np.int64)
    nsteps = (cin * k * k + 31) // 32
    live = st[:, 0] != 0
    b = st[live]
    nb = b.shape[0]
    rel = lambda i: (b[:, i] - b[:, 0]).mean()
    print(f"== {case}: {ms*1e3:.1f} us (events), {nb} stamped blocks, {nsteps} steps")
    print(f"   prologue issue {rel(1):.0f}  first stage landed {rel(2):.0f}  mainloop end {rel(3):.0f}  kernel end {rel(127):.0f} cycles since block entry")
    ns = min(nsteps, 40)
    vm = np.array([(b[:, 5 + 3 * s_] - b[:, 4 + 3 * s_]).mean() for s_ in range(ns)])
    bar = np.array([(b[:, 6 + 3 * s_] - b[:, 5 + 3 * s_]).mean() for s_ in range(ns)])
    per = np.array([(b[:, 6 + 3 * (s_ + 1)] - b[:, 6 + 3 * s_]).mean() for s_ in range(ns - 1)])
    print("   per step: period(avg) %.0f | vmcnt wait %.0f | barrier wait %.0f   (first 12 periods: %s)" % (per.mean() if len(per) else 0, vm.mean(), bar.mean(), " ".join("%.0f" % v for v in per[:12])))
    print("   vm waits:", " ".join("%.0f" % v for v in vm[:16]))
    print("   bar waits:", " ".join("%.0f" % v for v in bar[:16]))
