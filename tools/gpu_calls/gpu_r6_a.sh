#!/bin/bash
# round 6, first GPU run of the strip kernel (csrc/c3_tile.hip): its GPU tests, the per-launch parity of the C2 plan, bench A/B (strip kernel on / off), layer table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06a}
O=gpurun_out/$TAG
mkdir -p $O
date
timeout 600 python -m pytest tests/test_c3_fused_gpu.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "strip" > $O/pytest_strip.log 2>&1
echo "strip rc $?"; tail -15 $O/pytest_strip.log | cut -c1-400
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -s --timeout 800 -p no:cacheprovider -k "every_conv_launch and s_r60" > $O/pytest_parity.log 2>&1
echo "parity rc $?"; tail -12 $O/pytest_parity.log | cut -c1-300
date
timeout 600 python bench.py --config c2 > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-400 $O/bench_c2.json
YOLORT_AMD_C3_TILE=0 timeout 600 python bench.py --config c2 > $O/bench_c2_off.log 2>&1; grep '^{"metric' $O/bench_c2_off.log | tail -1 > $O/bench_c2_off.json; cut -c1-400 $O/bench_c2_off.json
date
for cfg in c2; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  tail -3 /tmp/ps_$cfg.log
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/rocprof_summary.py $db > $O/rocprof_summary_$cfg.csv 2>> $O/err.log
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
  grep "^# conv stack" $O/layer_table_$cfg.csv
  grep -i "tile\|C3" $O/layer_table_$cfg.csv | cut -c1-220
done
tail -5 $O/err.log
date
