#!/bin/bash
# round 3, call z9: conv3x3_res v4 (shortcut fetched first, DMA burst, epilogue steps in the last two thirds of the loop): tests, timeline, per-layer timings with / without a shortcut
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z9
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3_res" 2>&1 | tail -2 | tee gpurun_out/r03z9/tests.txt
(YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_r3stamps.so python tools/stamp_r3.py 8,64,64,320,320; RES=1 YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_r3stamps.so python tools/stamp_r3.py 8,64,64,320,320) 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|  d = \|print(" | tee gpurun_out/r03z9/stamps.txt
for lib in tools/_ab/lib_before_interleave.so tools/_ab/lib_interleave_v1.so tools/_ab/lib_interleave_v3.so yolort_amd/lib/libyolort_amd.so; do
for res in 0 1; do
echo "== $lib RES=$res" | tee -a gpurun_out/r03z9/conv_bench.txt
RES=$res YOLORT_AMD_LIB=$PWD/$lib TILES=93,132 timeout 300 python tools/conv_bench.py 32,64,64,80,80,3,1,1 8,64,64,320,320,3,1,1 64,64,48,320,320,3,1,1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03z9/conv_bench.txt
done; done
