#!/bin/bash
# round 6: timeline of the strip kernel (instrumented build, tools/build_c3t_stamps.sh) + the corrected per-launch table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06b}
O=gpurun_out/$TAG
mkdir -p $O
YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_c3tstamps.so timeout 300 python tools/stamp_c3t.py ${CASES:-32,40,40,256,128,1,0 32,40,40,256,128,3,1 32,80,80,128,64,2,1 32,80,80,256,64,1,0} > $O/stamps.txt 2>&1
cat $O/stamps.txt | cut -c1-260
for cfg in c2; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
  cut -c1-200 $O/layer_table_$cfg.csv
done
