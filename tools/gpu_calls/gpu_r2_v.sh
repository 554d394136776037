#!/bin/bash
# round-2 re-entry: the whole GPU suite at HEAD + the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -x -q --timeout 900 -p no:cacheprovider --durations=12 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log
timeout 300 python bench.py > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log > $O/bench_c2.json; python - <<'P'
import json
d=json.loads(open('gpurun_out/r02v/bench_c2.json').readline())
print('c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['frac_of_per_layer_bound'])
P
