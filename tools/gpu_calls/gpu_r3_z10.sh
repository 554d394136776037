#!/bin/bash
# round 3, call z10: shortcut loaded in packet form (two 16-byte loads per 32-channel group + lane unswap) in every lean epilogue: op tests, per-layer timings with a shortcut, same-box A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z10
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_c3_fused_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r03z10/tests.txt
for lib in tools/_ab/lib_before_interleave.so tools/_ab/lib_v4.so yolort_amd/lib/libyolort_amd.so; do
echo "== $lib RES=1" | tee -a gpurun_out/r03z10/conv_bench.txt
RES=1 YOLORT_AMD_LIB=$PWD/$lib TILES=93,132 timeout 300 python tools/conv_bench.py 32,64,64,80,80,3,1,1 8,64,64,320,320,3,1,1 64,64,48,320,320,3,1,1 32,128,128,40,40,3,1,1 32,256,256,20,20,3,1,1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03z10/conv_bench.txt
done
for rep in 1 2; do
for lib in tools/_ab/lib_before_interleave.so yolort_amd/lib/libyolort_amd.so; do
for cfg in c2 c5 c3; do
YOLORT_AMD_LIB=$PWD/$lib timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$cfg $lib', d['value'], d['ms_per_step'], 'conv serial', r['serial']['conv_ms_per_step'], 'frac', r['frac'])" | tee -a gpurun_out/r03z10/ab.txt
done; done; done
