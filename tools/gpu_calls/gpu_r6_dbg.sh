#!/bin/bash
# round 6: what bounds the strip kernel -- timings with one ingredient removed (tools/build_c3t_dbg.sh; results of these builds are garbage by construction)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06g}
O=gpurun_out/$TAG
mkdir -p $O
CASES=${CASES:-32,40,40,256,128,1,0 32,40,40,256,128,3,1 32,80,80,256,64,1,0}
echo "shipped:" > $O/dbg.txt
REPS=30 timeout 200 python tools/c3t_run.py $CASES 2>&1 | grep "^==" >> $O/dbg.txt
for n in 1 2 3 4 5; do
  echo "C3T_DBG=$n (1 no MFMA, 2 no DMA, 3 no SiLU math, 4 no barrier, 5 no fragment reads):" >> $O/dbg.txt
  YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_c3tdbg$n.so REPS=30 timeout 200 python tools/c3t_run.py $CASES 2>&1 | grep "^==" >> $O/dbg.txt
done
cat $O/dbg.txt
