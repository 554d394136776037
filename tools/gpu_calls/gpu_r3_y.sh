#!/bin/bash
# round 3, call y: conflict-free patch layout of the 8-wave halo kernel (pitch tw + 4, swizzle by tile index): op tests, then same-box A/B old / new library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03y
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo or 3x3 or conv" 2>&1 | tail -4 | tee gpurun_out/r03y/tests.txt
for rep in 1 2; do
for lib in tools/_ab/lib_old_patch.so yolort_amd/lib/libyolort_amd.so; do
for cfg in c2 c5; do
YOLORT_AMD_LIB=$PWD/$lib timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$cfg $lib', d['value'], d['ms_per_step'], 'conv serial', r['serial']['conv_ms_per_step'], 'frac', r['frac'])" | tee -a gpurun_out/r03y/ab.txt
done; done; done
