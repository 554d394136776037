"""Tuning aid: per-block s_memtime timeline of the 8-wave halo kernel (conv_halo8.hip).
build:  bash tools/build_h8_stamps.sh          (instrumented copy of conv_halo8 only, never the shipped library)
run:    YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_h8stamps.so python tools/stamp_h8.py n,cin,cout,h,w,tile [...]
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolort_amd import engine, _lib

dev = torch.device("cuda:0")
lib = _lib.load()
lib.ymi_debug_stamps_h8.restype = C.c_int
lib.ymi_debug_stamps_h8.argtypes = [C.c_void_p, C.c_int]

for case in sys.argv[1:]:
    n, cin, cout, h, w, tile = map(int, case.split(","))
    plan = engine.Plan(dev, torch.float16)
    x = plan.alloc(n, h, w, cin); x.base.normal_()
    wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    pc = engine.PackedConv(wt, None, None, torch.float16, dev)
    plan.conv(x, pc, 1, 1, tile=tile)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    ms = plan.profile(10)[0][1]
    plan.run(); torch.cuda.synchronize()
    st = np.zeros(2048 * 128, dtype=np.uint64)
    assert lib.ymi_debug_stamps_h8(st.ctypes.data, st.size) == 0
    st = st.reshape(2048, 128).astype(np.int64)
    nsteps = cin // 32 * 3
    b = st[st[:, 0] != 0]
    nblk = int(os.environ.get("NBLK", "0")) or b.shape[0]
    b = st[:nblk]                                      # rows of this launch's blocks (clocks of different XCDs are not comparable: no time filter)
    hw = b[:, 126]
    cu_key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)   # xcc, se, sh, cu
    per_cu = {}
    for k, t0_, t1_ in zip(cu_key, b[:, 0], b[:, 127]):
        per_cu.setdefault(int(k), []).append((int(t0_), int(t1_)))
    n_over = n_seq = 0
    for k, iv in per_cu.items():
        iv.sort()
        for (a0, a1), (b0, b1) in zip(iv, iv[1:]):
            if b0 < a1: n_over += 1
            else: n_seq += 1
    import collections
    rt = (b[:, 125] - b[:, 124]).astype(float)
    ok_rt = rt > 0
    print(f"   shader clock from s_memtime / s_memrealtime (100 MHz): {((b[ok_rt, 127] - b[ok_rt, 0]) / rt[ok_rt]).mean() * 100:.0f} MHz; block life {rt[ok_rt].mean() / 100:.2f} us; first start -> last end {(b[:, 125].max() - b[:, 124].min()) / 100:.2f} us")
    print(f"   {len(per_cu)} CUs used; blocks per CU histogram {sorted(collections.Counter(len(v) for v in per_cu.values()).items())}; consecutive block pairs on one CU: {n_over} overlapping, {n_seq} sequential")
    rel = lambda i: (b[:, i] - b[:, 0]).mean()
    t0 = b[:, 0].min()
    print(f"== {case}: {ms*1e3:.1f} us (events), {b.shape[0]} stamped blocks, {nsteps} steps; block entry spread {(b[:,0].max()-t0)} cyc, last exit {(b[:,127].max()-t0)} cyc")
    ent = np.sort(b[:, 0] - t0)
    print("   block entry times (cycles), deciles:", " ".join(str(int(ent[int(q * (len(ent) - 1))])) for q in np.linspace(0, 1, 11)))
    print(f"   setup {rel(1):.0f}  mainloop end {rel(3):.0f}  kernel end {rel(127):.0f} cycles since block entry (epilogue {rel(127)-rel(3):.0f})")
    ns = min(nsteps, 40)
    vm = np.array([(b[:, 4 + 3 * s_] - (b[:, 6 + 3 * (s_ - 1)] if s_ else b[:, 1])).mean() for s_ in range(ns)])
    bar = np.array([(b[:, 5 + 3 * s_] - b[:, 4 + 3 * s_]).mean() for s_ in range(ns)])
    comp = np.array([(b[:, 6 + 3 * s_] - b[:, 5 + 3 * s_]).mean() for s_ in range(ns)])
    print("   per step avg: vmcnt wait %.0f | barrier wait %.0f | issue+frag reads+MFMAs %.0f | period %.0f" % (vm[1:].mean(), bar[1:].mean(), comp[1:].mean(), (vm[1:] + bar[1:] + comp[1:]).mean()))
    print("   vm waits :", " ".join("%.0f" % v for v in vm[:14]))
    print("   bar waits:", " ".join("%.0f" % v for v in bar[:14]))
    print("   compute  :", " ".join("%.0f" % v for v in comp[:14]))
