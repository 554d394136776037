#!/bin/bash
# round 4, last call: the whole GPU suite + smoke at HEAD (after tiles 96 / 120 and their table entries), the C3 bench line and its rocprofv3 per-layer table again, the C2 line as a control
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04zzz
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -4 $O/pytest_all.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
for c in c3 c2; do
  timeout 600 python bench.py --config $c > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; cut -c1-200 $O/bench_$c.json
done
cfg=c3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
python tools/rocprof_summary.py $db > $O/rocprof_summary_$cfg.csv 2>> $O/err.log
python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
grep "^# conv stack" $O/layer_table_$cfg.csv
