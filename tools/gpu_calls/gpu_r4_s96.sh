#!/bin/bash
# round 4, call s96: yolov5m's 96 -> 96 1x1 at 320^2 (body.2.cv3, 746 us at 3.4 TB/s with tile 124) across the tiles that take it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04s96
DTYPE=bf16 TILES=124,123,122,121,21,12,111,112,142,141 timeout 300 python tools/conv_bench.py "64,96,96,320,320,1,1,0" "64,192,192,160,160,1,1,0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04s96/conv_bench.txt
