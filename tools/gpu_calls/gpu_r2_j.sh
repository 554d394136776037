#!/bin/bash
# round 2, GPU call J: anchor-split head, sort_image small-array fix, halo8 single patch buffer: tests + A/B + kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02j
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py tests/test_boundary_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $O/tests.log 2>&1; tail -5 $O/tests.log
for hs in 0 1 0 1; do
YOLORT_AMD_HEAD_SPLIT=$hs timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 > $O/hs_$hs.log 2>&1; python -c "
import json
d=json.loads([l for l in open('$O/hs_$hs.log') if l.startswith('{\"metric')][-1]); r=d['roofline']
print('head split $hs:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; conv excl', r['conv_ms_per_step'], 'post', r['other_kernels'])
"
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_c2.json > /tmp/ps_c2.log 2>&1)
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python tools/layer_table.py --ops $O/ops_c2.json --stats $db > $O/layer_table_c2.csv 2>> $O/err.log
python - <<'PY'
import csv
rows=[r for r in csv.reader(l for l in open('gpurun_out/r02j/layer_table_c2.csv') if not l.startswith('#'))]
ix={n:i for i,n in enumerate(rows[0])}
for r in rows[1:]:
    print(r[0], r[1][:36], r[ix['kernel']][:44], r[ix['avg_us']], r[ix['min_us']], r[ix['frac_of_own_bound']])
print(''.join(l for l in open('gpurun_out/r02j/layer_table_c2.csv') if l.startswith('#')))
PY
