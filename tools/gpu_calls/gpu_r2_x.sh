#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02x
mkdir -p $O
timeout 300 python -m pytest tests/test_boundary_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "gather" 2>&1 | tail -15
LB_COPY_REF=1 YOLORT_AMD_LETTERBOX=2 timeout 120 python tools/letterbox_bench.py c3 30 2>&1 | grep "^letterbox\|^reference" | tee $O/lb_debug.txt
for k in 2 4; do for d in 1 2 3; do
  YOLORT_AMD_LB_DEBUG=$d YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py c3 30 2>&1 | grep "^letterbox"
done; done | tee -a $O/lb_debug.txt
