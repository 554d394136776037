#!/bin/bash
# round 3, call z3 (deferred stores, prefetched shortcut): resident-weights persistent 3x3 for cin 48 / 64 (tile 132): op tests, per-layer timings against the tiles it replaces, then a same-box A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z3
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3_res" 2>&1 | tail -4 | tee gpurun_out/r03z3/tests.txt
TILES=93,92,113,132 timeout 300 python tools/conv_bench.py 32,64,64,80,80,3,1,1 8,64,64,320,320,3,1,1 64,64,48,320,320,3,1,1 64,48,48,320,320,3,1,1 32,64,64,160,160,3,1,1 2>&1 | tee gpurun_out/r03z3/conv_bench.txt
for rep in 1 2; do
for k in 0 1; do
for cfg in c2 c5 c3; do
YOLORT_AMD_RES3X3=$k timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$cfg res3x3=$k', d['value'], d['ms_per_step'], 'conv serial', r['serial']['conv_ms_per_step'], 'frac', r['frac'])" | tee -a gpurun_out/r03z3/ab.txt
done; done; done
