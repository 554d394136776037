#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "halo8 or halo" 2>&1 | tail -3
TILES=93,91,92,94,95 timeout 200 python tools/conv_bench.py 32,128,128,40,40,3,1,1 32,64,64,80,80,3,1,1 32,256,256,20,20,3,1,1 32,32,32,160,160,3,1,1 64,96,96,160,160,3,1,1 2>&1 | grep -v amdgpu.ids
