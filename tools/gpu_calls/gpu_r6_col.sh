#!/bin/bash
# round 6: column tiles of the strip kernel (maps wider than the LDS patch: yolov5l6 @ 1280) -- strip tests, the plans that now hold strip launches (C5), bench c5 / c2, per-launch table c5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06x}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_c3_fused_gpu.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "strip" > $O/pytest_strip.log 2>&1
echo "strip rc $?"; tail -4 $O/pytest_strip.log | cut -c1-300
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py tests/test_golden_gpu.py -m gpu -q -x --timeout 1200 -p no:cacheprovider > $O/pytest_parity.log 2>&1
echo "parity rc $?"; tail -4 $O/pytest_parity.log | cut -c1-300
for c in c5 c2; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; cut -c1-200 $O/bench_$c.json
done
for cfg in c5; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
  grep "c3_tile\|^# conv stack" $O/layer_table_$cfg.csv | cut -c1-220
done
