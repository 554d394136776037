#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ring
mkdir -p $O
timeout 400 python -m pytest tests/test_ops_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "pipelined_tiles or upsampled_second" 2>&1 | tail -5
TILES=0,111,115,116,112,117,118,119 timeout 300 python tools/conv_bench.py 32,512,512,20,20,1,1,0 32,256,256,20,20,1,1,0 32,256,256,40,40,1,1,0 32,128,128,40,40,1,1,0 32,512,256,40,40,1,1,0 32,128,256,80,80,3,2,1 32,256,512,40,40,3,2,1 32,1024,512,20,20,1,1,0 32,512,256,20,20,1,1,0 32,256,256,40,40,3,2,1 32,128,128,80,80,3,2,1 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_ring3.txt
