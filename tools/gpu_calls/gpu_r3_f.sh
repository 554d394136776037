#!/bin/bash
# round 3, call f: rocprofv3 per-layer table of C2 at HEAD (serial driver)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_c2.json > /tmp/ps_c2.log 2>&1)
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python tools/rocprof_summary.py $db > $O/rocprof_summary_c2.csv 2>> $O/err.log
python tools/layer_table.py --ops $O/ops_c2.json --stats $db > $O/layer_table_c2.csv 2>> $O/err.log
cut -c1-200 $O/layer_table_c2.csv; tail -3 $O/err.log
