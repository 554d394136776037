#!/bin/bash
# round 4, call h: ablation of tile 134 on yolov5s' body.3 and its 640^2 sibling: whole kernel / without the output stores (+0x800) / without the patch loads (+0x200) /
# patch traffic only (+0x1000)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04h
mkdir -p $O
TILES=134,2182,646,4230,2694 timeout 300 python tools/conv_bench.py 32,64,128,160,160,3,2,1 8,64,128,640,640,3,2,1 2>&1 | grep -v amdgpu.ids | tee $O/ablation_134.txt
