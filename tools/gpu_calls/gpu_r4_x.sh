#!/bin/bash
# round 4, call x: the general-tap 8-wave implicit GEMM (tiles 161-164, cin % 8 == 0) on yolov5m's 48-channel layers (bs 64 at 1280^2) against the 4-wave generic tiles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04x
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "general_tap" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests.txt
TILES=11,14,21,161,163 timeout 300 python tools/conv_bench.py "64,48,96,640,640,3,2,1" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
TILES=12,15,25,22,162,164 timeout 300 python tools/conv_bench.py "64,48,48,320,320,1,1,0" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
RES=1 TILES=12,15,25,162,164 timeout 300 python tools/conv_bench.py "64,48,48,320,320,3,1,1" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
TILES=11,12,21,161,162 timeout 300 python tools/conv_bench.py "32,16,32,320,320,3,2,1" "32,16,16,160,160,1,1,0" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
