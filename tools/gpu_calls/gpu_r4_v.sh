#!/bin/bash
# round 4, call v: eight further seeds of the spread workload (yolov5s; reference-exact goldens made by tests/golden/make_golden.py spread-more): the golden tests, then the
# default bench line with the pooled `parity.spread_more` block
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04v
mkdir -p $O
timeout 900 python -m pytest tests/test_golden_gpu.py -q -m gpu -k "spread" -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $O/golden_spread_verbose.txt
tail -5 $O/golden_spread_verbose.txt
( time timeout 900 python bench.py ) > $O/bench_c2.json 2> $O/bench_c2.err
tail -3 $O/bench_c2.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04v/bench_c2.json') if l.startswith('{"metric')][0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['parity'].get('spread_more'), indent=1)[:3000])
print('unexplained', d['parity'].get('unexplained'))
PY
