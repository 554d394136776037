#!/bin/bash
# round 4, call h96: conv_halo8 in 96-cout blocks (tile 96) on yolov5m's 3x3 shapes (bs 64 at 1280^2, bf16) against the 128-wide tiles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04h96
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "96_cout" -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
export DTYPE=bf16
TILES=91,95,31,34,96 timeout 300 python tools/conv_bench.py "64,96,96,160,160,3,1,1" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
RES=1 TILES=91,95,31,34,96 timeout 300 python tools/conv_bench.py "64,96,96,160,160,3,1,1" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
TILES=91,93,95,92,96 timeout 300 python tools/conv_bench.py "64,192,192,80,80,3,1,1" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
RES=1 TILES=91,93,95,92,96 timeout 300 python tools/conv_bench.py "64,192,192,80,80,3,1,1" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
