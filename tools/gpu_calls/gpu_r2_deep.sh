#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "stream or chained" 2>&1 | tail -3
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
}
for rep in 1 2 3; do
run "two-deep ring (old)" YOLORT_AMD_STREAM_DEEP=0
run "three-deep ring" A=1
done
for d in 0 1; do
YOLORT_AMD_STREAM_DEEP=$d TILES=121,122 timeout 200 python tools/conv_bench.py 32,64,64,160,160,1,1,0 32,64,64,80,80,1,1,0 2>&1 | grep -v amdgpu.ids | sed "s/^/deep $d: /"
done
