#!/bin/bash
# round 3, call x: fused stem + body.1 from the letterboxed canvas (dynamic-shape streams): parity, then same-box A/B on the c2dyn stream
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03x
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_c3_fused_gpu.py tests/test_cabi.py -x -q -m gpu -k "launch or stem or abi" -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r03x/tests.txt
tail -5 gpurun_out/r03x/tests.txt
for rep in 1 2; do
for k in 0 1; do
YOLORT_AMD_FUSE_STEM_CANVAS=$k timeout 400 python bench.py --config c2 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; y=r['other_kernels']['dynamic_shape_stream']; print('c2 canvas_fuse=$k', d['value'], r['serial']['conv_ms_per_step'], 'dyn conv serial', y['conv_ms_per_step_serial'], y['letterbox_tile2_kernel']['ms'])" | tee -a gpurun_out/r03x/ab.txt
done; done
