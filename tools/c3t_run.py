"""Tuning aid: runs C3 blocks through the plan N times (for rocprofv3 --pmc / --kernel-trace of the strip kernel, csrc/c3_tile.hip).
    python tools/c3t_run.py n,h,w,c_in,hidden,bottlenecks,shortcut [...]    (env REPS, default 20)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolort_amd import engine
from yolort_amd.v5.models.common import C3

dev = torch.device("cuda:0")
reps = int(os.environ.get("REPS", "20"))
for case in sys.argv[1:]:
    n, h, w, c1, c_, nb, sc = map(int, case.split(","))
    m = C3(c1, 2 * c_, n=nb, shortcut=bool(sc)).eval()
    plan = engine.Plan(dev, torch.float16)
    x = plan.alloc(n, h, w, c1); x.base.normal_()
    m.emit(plan, x, name="c3")
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    prof = plan.profile(reps)
    print(f"== {case}: " + ", ".join(f"{nm} {ms * 1e3:.1f} us" for nm, ms, _ in prof), flush=True)
