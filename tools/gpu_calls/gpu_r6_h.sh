#!/bin/bash
# round 6: strip kernel iteration -- GPU tests, single-block timings, bench, per-launch table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06h}
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python -m pytest tests/test_c3_fused_gpu.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "strip" > $O/pytest_strip.log 2>&1
echo "strip rc $?"; tail -5 $O/pytest_strip.log | cut -c1-400
REPS=30 timeout 200 python tools/c3t_run.py 32,40,40,256,128,1,0 32,40,40,512,128,1,0 32,40,40,256,128,3,1 32,80,80,256,64,1,0 32,80,80,128,64,2,1 2>&1 | grep "^==" | tee $O/c3t_run.txt
timeout 600 python bench.py --config c2 > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-200 $O/bench_c2.json
for cfg in c2; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
  grep "c3_tile\|^# conv stack" $O/layer_table_$cfg.csv | cut -c1-200
done
