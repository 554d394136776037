#!/bin/bash
# round 4, call y: the whole GPU suite at HEAD (after the SPP block-size change, the letterbox variants, the further spread seeds)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04y
mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 | cut -c1-300 ) 2>&1 | tee $O/pytest_gpu.txt
