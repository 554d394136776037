"""Tuning / debugging aid: reproduces tests/test_configs_gpu.py::test_config5 on the GPU and dumps what a CPU session needs to look at a disagreement between the HIP
post-process and the oracle's on one image: the unfused fp32 logits of that image and the HIP detections (gpurun_out/<dir>/c5_case.npz)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import yolov5_oracle as O  # noqa: E402
from yolort_amd.models import YOLOv5  # noqa: E402
from workloads.synth import synth_images, synth_weights  # noqa: E402

out = sys.argv[1]
dev = torch.device("cuda:0")
arch, thr = "yolov5_darknet_pan_l6_r60", 0.25
m = YOLOv5(arch=arch, size=(1280, 1280), size_divisible=64, score_thresh=thr, nms_thresh=0.45, detections_per_img=300)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=3.0))
m = m.to(dev).half().eval()
imgs = [im.to(dev).half() for im in synth_images(8, 1280, 1280, seed=1)]
dets = m.predict(imgs)
m.model.fuse_head_decode = False
dets_u = m.predict(imgs)
m.model.post_exact_full = True
dets_f = m.predict(imgs)
e = next(iter(m.model._entries.values()))
logits = [v.as_tensor().cpu().view(v.n, v.h, v.w, 3, 85).permute(0, 3, 1, 2, 4).contiguous() for v in e.logits]
strides, anchors = O.anchors_for(4)
with torch.no_grad():
    pred = O.decode(logits, strides, anchors)
    ref = O.postprocess(pred, thr, 0.45, 300)
bad = [i for i in range(8) if not np.array_equal(dets[i]["labels"].cpu().numpy(), ref[i]["labels"].numpy())]
print("images whose label sequences differ from the oracle's:", bad, "| prefix vs exact-full identical:",
      all(torch.equal(a[k], b[k]) for a, b in zip(dets_u, dets_f) for k in a))
save = {}
for i in bad[:1]:
    for l, lg in enumerate(logits):
        save[f"logits{l}"] = lg[i].numpy().astype(np.float32)
    for k in ("boxes", "scores", "labels"):
        save[f"hip_{k}"] = dets[i][k].float().cpu().numpy() if k != "labels" else dets[i][k].cpu().numpy()
        save[f"ref_{k}"] = ref[i][k].numpy()
    save["image"] = np.int32(i)
if save:
    np.savez_compressed(os.path.join(out, "c5_case.npz"), **save)
    print("saved", os.path.getsize(os.path.join(out, "c5_case.npz")) // 2**20, "MiB")
