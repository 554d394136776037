import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from yolort_amd.engine import Plan, View
from oracle import yolov5_oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
n, nc = 3, 80
shapes = [(20, 24), (10, 12), (5, 6)]
heads = [torch.randn(n, 3, h, w, nc + 5, generator=g) * 2.0 - 1.0 for h, w in shapes]
strides, anchors = O.anchors_for(3)
k = nc + 5
for cap in (6144, 16384 * 3):
    plan = Plan(dev, torch.float16)
    ins = []
    for ho in heads:
        nhwc = torch.zeros(n, ho.shape[2], ho.shape[3], 256, device=dev)
        nhwc[..., :255] = ho.to(dev).permute(0, 2, 3, 1, 4).reshape(n, ho.shape[2], ho.shape[3], 255)
        ins.append(nhwc)
    views = [View(t.view(-1), 0, n, t.shape[1], t.shape[2], 255, 256) for t in ins]
    pb = plan.postprocess(views, strides, anchors, nc, 0.3, 0.45, 300, cap)
    plan.run(); torch.cuda.synchronize()
    print("cap", cap, "status", pb.status.cpu().tolist(), "count", pb.count.cpu().tolist())
