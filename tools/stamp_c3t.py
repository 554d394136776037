"""Tuning aid: per-wave s_memtime timeline of the strip kernel (csrc/c3_tile.hip).
build:  bash tools/build_c3t_stamps.sh          (instrumented copy of c3_tile.hip only, never the shipped library)
run:    YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_c3tstamps.so python tools/stamp_c3t.py n,h,w,c_in,hidden,bottlenecks,shortcut [...]
Records: 0 kernel entry, 8 tile start, 1 step wait begin, 2 own pieces landed, 3 barrier passed, 4 step compute done, 9 / 10 / 11 phase B / C / D begin, 12 tile end.
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolort_amd import engine, _lib
from yolort_amd.v5.models.common import C3

dev = torch.device("cuda:0")
lib = _lib.load()
NST = 256
lib.ymi_debug_stamps_c3t.restype = C.c_int
lib.ymi_debug_stamps_c3t.argtypes = [C.c_void_p, C.c_int]

for case in sys.argv[1:]:
    n, h, w, c1, c_, nb, sc = map(int, case.split(","))
    m = C3(c1, 2 * c_, n=nb, shortcut=bool(sc)).eval()
    plan = engine.Plan(dev, torch.float16)
    x = plan.alloc(n, h, w, c1); x.base.normal_()
    m.emit(plan, x, name="c3")
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    prof = plan.profile(10)
    print(f"== {case}: " + ", ".join(f"{nm} {ms * 1e3:.1f} us" for nm, ms, _ in prof))
    for op in range(plan.num_ops):
        lib.ymi_debug_stamps_c3t_clear()
        plan.run(op, op + 1); torch.cuda.synchronize()
        st = np.zeros(256 * 8 * NST, dtype=np.uint64)
        assert lib.ymi_debug_stamps_c3t(st.ctypes.data, st.size) == 0
        st = st.reshape(256, 8, NST)
        ids = (st >> np.uint64(56)).astype(np.int64)
        tm = (st & np.uint64((1 << 56) - 1)).astype(np.int64)
        used = tm[:, 0, 0] != 0
        nblk = int(used.sum())
        print(f"-- op {op} {plan.names[op]}: {nblk} stamped blocks")
        for wv in (0, 1, 4, 7):
            I, T = ids[used, wv], tm[used, wv]
            t0 = T[:, 0:1]
            # per record position: id (same for every block) and mean time since kernel entry
            cnt = int((T[0] != 0).sum())
            line, prev = [], 0
            phase = "A"
            waits, bars, comps, gaps = {}, {}, {}, {}
            last4 = None
            for k in range(cnt):
                i = int(I[0, k]); t = float((T[:, k] - t0[:, 0]).mean())
                if i in (9, 10, 11): phase = {9: "B", 10: "C", 11: "D"}[i]
                if i == 8: phase = "A"
                if i == 1:
                    t1 = t
                    if last4 is not None: gaps.setdefault(phase, []).append(t - last4)
                if i == 2: waits.setdefault(phase, []).append(t - t1); t2 = t
                if i == 3: bars.setdefault(phase, []).append(t - t2); t3 = t
                if i == 4: comps.setdefault(phase, []).append(t - t3); last4 = t
                if i in (8, 9, 10, 11, 12): line.append(f"{ {8:'tile',9:'B',10:'C',11:'D',12:'end'}[i] }@{t:.0f}")
            print(f"   wave {wv}: " + " ".join(line))
            for ph in "ABCD":
                if ph in comps:
                    f = lambda d: f"{np.mean(d[ph]):.0f}" if ph in d else "-"
                    print(f"      phase {ph}: {len(comps[ph])} steps; per step: vmcnt wait {f(waits)}  barrier {f(bars)}  compute {f(comps)}  gap before {f(gaps)}   (first step wait {waits[ph][0]:.0f} bar {bars[ph][0]:.0f})")
