"""How reproducible is the fp32 CPU reference itself?  The oracle (fp32) against its own float64 evaluation on the 32 images of
BASELINE configs[1] (yolov5s, head gain 0.4, thr 0.25), scored with bench.direct_checks -- the yardstick for the fp32 parity mode
of the HIP path (tests/test_parity_gpu.py).  CPU only (test infrastructure).  Measured (8 threads, 14 s):
  IoU >= 0.999: paired 6310 of 6445, 268 unpaired, equal count in 30/32 images, min IoU of pairs 0.99911, max |dscore| 3.9e-5
  IoU >= 0.99 : paired 6415 of 6445,  58 unpaired
"""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import yolov5_oracle as O
from yolort_amd.models import YOLOv5
from workloads.synth import synth_images, synth_weights
import bench
torch.set_num_threads(8)
arch="yolov5_darknet_pan_s_r60"; thr=0.25
m=YOLOv5(arch=arch, score_thresh=thr)
sd=synth_weights(m.state_dict(), arch, seed=0, head_gain=0.4)
imgs=[synth_images(1,640,640,seed=i+1)[0] for i in range(32)]
npd=lambda d: {"boxes":d["boxes"].numpy(),"scores":d["scores"].numpy(),"labels":d["labels"].numpy()}
sd64={k:(v.double() if v.is_floating_point() else v) for k,v in sd.items()}
R32=[];R64=[]
t=time.time()
with torch.no_grad():
    for i in range(0,32,4):
        batch,_=O.letterbox(imgs[i:i+4])
        d32=O.yolo_forward(batch, sd, thr, 0.45, 300, p="model.")
        f64=O.backbone(batch.double(), sd64, "model.backbone")
        ho=O.head(f64, sd64, "model.head")
        s,a=O.anchors_for(3)
        d64=O.postprocess(O.decode([h.float() for h in ho], s, a), thr, 0.45, 300)
        R32+= [npd(d) for d in d32]; R64+=[npd(d) for d in d64]
print("time",time.time()-t)
for iou_min in (1-1e-3, 0.99):
    print(iou_min, bench.direct_checks(R32,R64,thr,iou_min=iou_min))
