#!/bin/bash
# round 6: same-box comparison of several strip-kernel builds (tools/_bin/libyolort_amd_<alt>.so for alt in $ALTS) against the shipped library, single launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06multi}
O=gpurun_out/$TAG
mkdir -p $O
CASES=${CASES:-32,40,40,256,128,1,0 32,40,40,256,128,3,1 32,80,80,256,64,1,0 32,80,80,128,64,2,1}
for rep in 1 2; do
  echo "shipped:" >> $O/multi.txt
  REPS=30 timeout 200 python tools/c3t_run.py $CASES 2>&1 | grep "^==" >> $O/multi.txt
  for alt in $ALTS; do
    echo "$alt:" >> $O/multi.txt
    YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_$alt.so REPS=30 timeout 200 python tools/c3t_run.py $CASES 2>&1 | grep "^==" >> $O/multi.txt
  done
done
cat $O/multi.txt
