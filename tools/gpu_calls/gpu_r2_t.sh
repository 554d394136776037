#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "c32" 2>&1 | tail -5
TILES=0,131,94,93,62,115 timeout 200 python tools/conv_bench.py 32,32,32,160,160,3,1,1 32,32,64,320,320,3,2,1 2>&1 | grep -v amdgpu.ids
