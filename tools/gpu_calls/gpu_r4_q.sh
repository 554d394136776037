#!/bin/bash
# round 4, call q: the spread_m golden in bf16: what the HIP path pairs under the generous pairing (same label, IoU >= 0.5, |dscore| <= 0.1) next to the reference's own bf16 run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04q
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04q/spread_m_bf16.txt
import json, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from test_golden_gpu import _golden, _model, _np, _ref16
from workloads.synth import spread_images
dev = torch.device('cuda:0')
meta, ref, _ = _golden('spread', 'm')
imgs = spread_images(meta['arch'], meta['seed'])
for dtype in (torch.bfloat16, torch.float16):
    m = _model(meta, dev, dtype, 'spread')
    got = [_np(d) for d in m.predict([im.to(dev).to(dtype) for im in imgs])]
    for eps, iou in ((0.06, 0.9), (0.1, 0.5), (0.2, 0.5), (0.5, 0.3)):
        print(dtype, 'eps', eps, 'iou', iou, bench.direct_checks(ref, got, meta['thr'], score_eps=eps, iou_min=iou))
    print('own', _ref16('spread', 'm', dtype))
    del m
PY
