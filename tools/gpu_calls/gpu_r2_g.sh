#!/bin/bash
# round 2, GPU call G: A/B of tile tables on one box (e2e throughput), chain diagnostics, per-layer table with PMC
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
date
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -s --timeout 300 -p no:cacheprovider -k "chained_1x1" > $O/tests_chain.log 2>&1; grep -E "differ|passed|failed" $O/tests_chain.log | head -12
for rep in 1 2; do
for tab in r02a r02f; do
YOLORT_AMD_TILE_TABLE_PATH=$PWD/tools/_ab/$tab.json timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 > $O/ab_$tab.log 2>&1; python -c "
import json
d=json.loads([l for l in open('$O/ab_$tab.log') if l.startswith('{\"metric')][-1]); r=d['roofline']
print('$tab rep$rep', d['value'], 'img/s', d['ms_per_step'], 'ms/step; conv excl', r['conv_ms_per_step'], 'in-region', r['in_timed_region']['conv_ms_per_step'])
"
done
done
date
