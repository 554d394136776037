#!/bin/bash
# round 6: strip geometries measured against the cost model's choice (YOLORT_AMD_C3T_GEOM = "R,delta,column tiles"; single launches, tools/c3t_run.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06geom}
O=gpurun_out/$TAG
mkdir -p $O
C64="32,80,80,256,64,1,0 32,80,80,128,64,2,1"
C128="32,40,40,256,128,1,0 32,40,40,256,128,3,1"
for g in default 10,1,2 10,11,2 6,15,1 14,1,2 5,1,1 4,1,1; do
  echo "hidden 64 @ 80x80, GEOM $g:" >> $O/geom.txt
  if [ $g = default ]; then REPS=30 timeout 200 python tools/c3t_run.py $C64 2>&1 | grep "^==" >> $O/geom.txt
  else YOLORT_AMD_C3T_GEOM=$g REPS=30 timeout 200 python tools/c3t_run.py $C64 2>&1 | grep "^==\|rror" | head -4 >> $O/geom.txt; fi
done
for g in default 10,1,2 10,5,2 5,1,1 20,1,4 4,9,1; do
  echo "hidden 128 @ 40x40, GEOM $g:" >> $O/geom.txt
  if [ $g = default ]; then REPS=30 timeout 200 python tools/c3t_run.py $C128 2>&1 | grep "^==" >> $O/geom.txt
  else YOLORT_AMD_C3T_GEOM=$g REPS=30 timeout 200 python tools/c3t_run.py $C128 2>&1 | grep "^==\|rror" | head -4 >> $O/geom.txt; fi
done
cat $O/geom.txt
