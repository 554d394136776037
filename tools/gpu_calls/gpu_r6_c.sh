#!/bin/bash
# round 6: strip kernel iteration -- GPU tests, timeline (instrumented build), bench, per-launch table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06c}
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python -m pytest tests/test_c3_fused_gpu.py -m gpu -q -x --timeout 500 -p no:cacheprovider -k "strip" > $O/pytest_strip.log 2>&1
echo "strip rc $?"; tail -5 $O/pytest_strip.log | cut -c1-400
YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_c3tstamps.so timeout 300 python tools/stamp_c3t.py ${CASES:-32,40,40,256,128,1,0 32,80,80,256,64,1,0} > $O/stamps.txt 2>&1
grep -v "wave 1:\|wave 7:" $O/stamps.txt | cut -c1-230
timeout 600 python bench.py --config c2 > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-200 $O/bench_c2.json
for cfg in c2; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
  grep "c3_tile\|^# conv stack" $O/layer_table_$cfg.csv | cut -c1-200
done
