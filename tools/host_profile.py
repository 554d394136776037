"""Tuning aid: cProfile of the host side of forward_async / result on the C2 workload (where do the ~0.76 ms per batch go?)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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


def loop(n=200):
    pend = []
    for _ in range(n):
        pend.append(m.forward_async(imgs))
        if len(pend) > 3:
            pend.pop(0).result()
    while pend:
        pend.pop(0).result()


pr = cProfile.Profile()
pr.enable()
loop()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(45)
