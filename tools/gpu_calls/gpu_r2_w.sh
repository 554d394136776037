#!/bin/bash
# lean letterbox kernel: bit-equality tests, gather redo test, C3 at spec, kernel sweep
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02w
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_boundary_gpu.py tests/test_configs_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "letterbox or gather or config3" 2>&1 | tail -6
for k in tile1 1 2 4; do
  YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py c3 30 2>&1 | grep "^letterbox"
done | tee $O/letterbox_sweep.txt
for k in tile1 1 2 4; do
  YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py c2dyn 50 2>&1 | grep "^letterbox"
done | tee -a $O/letterbox_sweep.txt
