"""Times the first layers of the yolov5s bs-32 640x640 plan in isolation (HIP events, one stream): the planar stem + body.1 as two launches
against the fused launch (csrc/stem_body1_fused.hip), and the fused C3 that follows.  python tools/stem_bench.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yolort_amd.models import YOLOv5  # noqa: E402
from workloads.synth import synth_images, synth_weights  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
arch = "yolov5_darknet_pan_s_r60"
m = YOLOv5(arch=arch, size=(640, 640), score_thresh=0.25)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=0.4))
m = m.to(dev).half().eval()
imgs = [im.to(dev).half() for im in synth_images(32, 640, 640, seed=1)]
m.predict(imgs)
torch.cuda.synchronize()
plan = next(iter(m.model._entries.values())).plan


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def two():
    os.environ["YOLORT_AMD_FUSE_STEM"] = "0"
    plan.stem_from_planar(imgs)
    plan.run(1, 2)


def fused():
    os.environ["YOLORT_AMD_FUSE_STEM"] = "1"
    plan.stem_from_planar(imgs)


print(f"stem + body.1, two launches: {timed(two):.1f} us   fused: {timed(fused):.1f} us   op 2 ({plan.names[2]}): {timed(lambda: plan.run(2, 3)):.1f} us", flush=True)
