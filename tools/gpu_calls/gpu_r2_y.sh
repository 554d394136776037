#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02y
mkdir -p $O
timeout 300 python -m pytest tests/test_boundary_gpu.py tests/test_ops_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "gather or letterbox" 2>&1 | tail -8
for cg in 1 2; do for k in 2 4; do for d in 0 4; do
  YOLORT_AMD_LB_CG=$cg YOLORT_AMD_LB_DEBUG=$d YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py c3 30 2>&1 | grep "^letterbox" | sed "s/^letterbox/letterbox cg=$cg/"
done; done; done | tee $O/lb_sweep2.txt
YOLORT_AMD_LB_CG=2 YOLORT_AMD_LB_DEBUG=2 YOLORT_AMD_LETTERBOX=2 timeout 120 python tools/letterbox_bench.py c3 30 2>&1 | grep "^letterbox" | sed "s/^letterbox/letterbox cg=2/" | tee -a $O/lb_sweep2.txt
for cg in 1 2; do YOLORT_AMD_LB_CG=$cg YOLORT_AMD_LETTERBOX=2 timeout 120 python tools/letterbox_bench.py c2dyn 50 2>&1 | grep "^letterbox" | sed "s/^letterbox/letterbox cg=$cg/"; done | tee -a $O/lb_sweep2.txt
