"""Tuning aid: host-side cost of one forward_async() submit and one result() collection (yolov5s bs 32, planar stem path, graph replay = the default),
with the weights validated per batch (default) and frozen (`freeze_weights()`), and of `weights_signature` alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolort_amd import hipmodule
from yolort_amd.models import YOLOv5
from workloads.synth import synth_images, synth_weights

dev = torch.device("cuda:0")
arch = "yolov5_darknet_pan_s_r60"
m = YOLOv5(arch=arch, size=(640, 640), score_thresh=0.25)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=0.4))
m = m.to(dev).half().eval()
imgs = [im.to(dev).half() for im in synth_images(32, 640, 640, seed=1)]
for _ in range(8):
    m.forward_async(imgs).result()
torch.cuda.synchronize()


def submit_cost(n=300):
    """one batch at a time, the GPU idle at every submit: pure host cost of the submit, then of the collection"""
    ts, tc = 0.0, 0.0
    for _ in range(n):
        t0 = time.perf_counter()
        p = m.forward_async(imgs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        p.result()
        t3 = time.perf_counter()
        ts += t1 - t0
        tc += t3 - t2
    return 1e3 * ts / n, 1e3 * tc / n


print("signature extension loaded:", getattr(hipmodule, "_SIG_EXT", None) is not None)
t0 = time.perf_counter()
for _ in range(2000):
    hipmodule.weights_signature(m.model)
print(f"weights_signature: {1e3 * (time.perf_counter() - t0) / 2000:.4f} ms per call")
s, c = submit_cost()
print(f"default: submit {s:.4f} ms per batch; collect (GPU already idle) {c:.4f} ms per batch")
m.freeze_weights()
s, c = submit_cost()
print(f"frozen : submit {s:.4f} ms per batch; collect {c:.4f} ms per batch")
m.freeze_weights(False)

# where the submit's time goes: each C entry point of the path behind a timer (the wrappers add ~0.3 us per call)
lib = next(iter(m.model._entries.values())).plan.lib
acc = {}


def timed(name):
    f = getattr(lib, name)

    def g(*a):
        t0 = time.perf_counter()
        r = f(*a)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return g


class _Wrapped:
    def __init__(self, lib):
        self._lib = lib
        self._w = {n: timed(n) for n in ("ymi_plan_begin", "ymi_plan_submit", "ymi_stem_body1_planar", "ymi_conv_stem_planar", "ymi_plan_done_query", "ymi_plan_done_sync", "ymi_plan_run")}

    def __getattr__(self, n):
        return self._w.get(n) or getattr(self._lib, n)


w = _Wrapped(lib)
for e in [en for r in m.model._ring.values() for en in r]:
    e.plan.lib = w
    if e.c_done is not None:
        e.c_done.lib = w
n = 300
s, c = submit_cost(n)
print(f"default, C entry points timed: submit {s:.4f} ms per batch, of which " + ", ".join(f"{k} {1e3 * v / n:.4f}" for k, v in sorted(acc.items())))
