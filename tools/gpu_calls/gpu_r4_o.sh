#!/bin/bash
# round 4, call o: the whole GPU suite at HEAD (re-tuned table, serving mode, row-store upsampled copy, C++ op registration) + smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04o
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -12 $O/pytest_all.log | cut -c1-400
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
