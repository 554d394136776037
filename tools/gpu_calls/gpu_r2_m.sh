#!/bin/bash
# round 2, GPU call M: lean epilogue + bias behind DMA + pointwise geometry + head as 1x1 + ranking sort + persistent stem: full suite, bench, trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02m
mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q -x --timeout 900 -p no:cacheprovider > $O/tests.log 2>&1; tail -5 $O/tests.log
for i in 1 2; do
timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 > $O/bench_c2_$i.log 2>&1; grep '^{"metric' $O/bench_c2_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['other_kernels'])"
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_c2.json > /tmp/ps_c2.log 2>&1)
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python tools/layer_table.py --ops $O/ops_c2.json --stats $db > $O/layer_table_c2.csv 2>> $O/err.log
python - <<'PY'
import csv
rows=[r for r in csv.reader(l for l in open('gpurun_out/r02m/layer_table_c2.csv') if not l.startswith('#'))]
ix={n:i for i,n in enumerate(rows[0])}
for r in rows[1:]:
    print(r[0], r[1][:36], r[ix['kernel']][:44], r[ix['avg_us']], r[ix['frac_of_own_bound']])
print(''.join(l for l in open('gpurun_out/r02m/layer_table_c2.csv') if l.startswith('#')))
PY
