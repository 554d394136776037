#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02z
mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "letterbox" 2>&1 | tail -4
for w in c3 c2dyn; do
YOLORT_AMD_LETTERBOX=tile1 timeout 120 python tools/letterbox_bench.py $w 30 2>&1 | grep "^letterbox"
for cg in 1 2; do for k in 1 2 4; do for d in 0 4; do
  YOLORT_AMD_LB_CG=$cg YOLORT_AMD_LB_DEBUG=$d YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py $w 30 2>&1 | grep "^letterbox" | sed "s/^letterbox/letterbox cg=$cg/"
done; done; done; done | tee $O/lb_sweep3.txt
