#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "stream or chained or second_output" 2>&1 | tail -3
for p in 0 1; do
YOLORT_AMD_STREAM_TP=$p timeout 100 python tools/stream_chain_bench.py 64 64 160 160 122,121 8 2>&1 | grep "^conv" | sed "s/^/row stores $p: /"
done
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
}
for rep in 1 2 3; do
run "packet stores (old)" YOLORT_AMD_STREAM_TP=0
run "row stores" A=1
done
