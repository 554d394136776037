#!/bin/bash
# round 4, call r: the yolov5l6 golden (conditioned recipe, threshold in a gap): fp32 parity mode and fp16 path; the tests that failed in call o after their fixes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04r
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04r/cond_l6.txt
import json, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from test_golden_gpu import _golden, _model, _np
from workloads.synth import cond_images
dev = torch.device('cuda:0')
meta, ref, _ = _golden('cond', 'l6')
imgs = cond_images(meta['arch'], meta['seed'])
for dtype in (torch.float32, torch.float16):
    m = _model(meta, dev, dtype, 'cond')
    got = [_np(d) for d in m.predict([im.to(dev) if dtype == torch.float32 else im.to(dev).to(dtype) for im in imgs])]
    for eps, iou in ((1e-4, 1 - 1e-3), (1e-2, 0.98), (0.1, 0.5)):
        print(dtype, 'eps', eps, 'iou', iou, bench.direct_checks(ref, got, meta['thr'], score_eps=eps, iou_min=iou))
    del m
PY
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_golden_gpu.py tests/test_boundary_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | cut -c1-300 | tee gpurun_out/r04r/tests.txt
