#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
}
for rep in 1 2 3; do
run "grid 4/CU (old)" YOLORT_AMD_STREAM_BLOCKS=4
run "grid = resident (new)" A=1
done
for b in 4 3 2; do
YOLORT_AMD_STREAM_BLOCKS=$b TILES=121,122 timeout 200 python tools/conv_bench.py 32,64,64,160,160,1,1,0 32,64,64,80,80,1,1,0 2>&1 | grep -v amdgpu.ids | sed "s/^/blocks $b: /"
done
