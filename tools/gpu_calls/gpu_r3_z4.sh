#!/bin/bash
# round 3, call z4: where a tile's time goes in conv3x3_res (debug bits: 1 = no MFMA loop, 2 = no epilogue, 4 = no patch prefetch after the first, 8 = weight fragments read once, 16 = activation fragments read once)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z4
TILES=132,2180,4228,6276,2692,6788 timeout 300 python tools/conv_bench.py 8,64,64,320,320,3,1,1 2>&1 | tee -a gpurun_out/r03z4/conv_bench.txt
