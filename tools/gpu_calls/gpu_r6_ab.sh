#!/bin/bash
# round 6: same-box A/B of a strip-kernel build knob -- the shipped library against tools/_bin/libyolort_amd_$ALT.so (single launches + the serial per-launch table)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06ab}
ALT=${ALT:-nospread}
O=gpurun_out/$TAG
mkdir -p $O
CASES=${CASES:-32,40,40,256,128,1,0 32,40,40,512,128,1,0 32,40,40,256,128,3,1 32,80,80,256,64,1,0 32,80,80,128,64,2,1}
for rep in 1 2; do
  echo "shipped:" >> $O/ab.txt
  REPS=30 timeout 200 python tools/c3t_run.py $CASES 2>&1 | grep "^==" >> $O/ab.txt
  echo "$ALT:" >> $O/ab.txt
  YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_$ALT.so REPS=30 timeout 200 python tools/c3t_run.py $CASES 2>&1 | grep "^==" >> $O/ab.txt
done
cat $O/ab.txt
timeout 600 python -m pytest tests/test_c3_fused_gpu.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "strip" 2>&1 | tail -2
