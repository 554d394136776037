#!/bin/bash
# round 2, GPU call K: gather test diagnostics, halo8 timelines, head with / without candidates, persistent stem
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
timeout 600 python -m pytest tests/test_boundary_gpu.py tests/test_ops_gpu.py tests/test_e2e_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gather or stem or planar or e2e" > $O/tests.log 2>&1; tail -3 $O/tests.log; grep -n "mismatch image\|total detections" $O/tests.log | head
YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_h8stamps.so timeout 300 python tools/stamp_h8.py 32,32,32,160,160,94 32,128,128,40,40,93 32,64,64,80,80,93 32,256,256,20,20,92 > $O/stamps.txt 2>&1; cat $O/stamps.txt
for thr in 0.25 0.999; do
(cd /tmp && rm -rf /tmp/prof_t && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 --score-thresh $thr --ops $GRAFT_REPO_ROOT/$O/ops_c2.json > /tmp/ps_t.log 2>&1)
db=$(find /tmp/prof_t -name "*.db" | head -1)
python tools/layer_table.py --ops $O/ops_c2.json --stats $db > $O/layer_table_c2_thr$thr.csv 2>> $O/err.log
echo "== thr $thr"; grep -E "^(0|1|2|3|47)," $O/layer_table_c2_thr$thr.csv | cut -d, -f1,2,5-7 | cut -c1-150; grep "^#" $O/layer_table_c2_thr$thr.csv | cut -c1-120
done
timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
