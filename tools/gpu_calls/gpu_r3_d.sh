#!/bin/bash
# round 3, call d: re-tune the whole pinned tile table (all four default configs, TP tiles offered), A/B against the committed table; new bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 2500 $O/bench_c2.json; tail -3 $O/bench_c2.err
timeout 900 python tools/tune_tiles.py --out $O/tiles_new.json > $O/tune.log 2>&1; tail -3 $O/tune.log | cut -c1-200
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: $cfg', d['value'], d['ms_per_step'], d['roofline']['serial']['conv_ms_per_step'], d['roofline']['frac'])"
}
for cfg in c2 c3 c5; do
for rep in 1 2; do
run "committed table" $cfg A=1
run "re-tuned table" $cfg YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_new.json
done; done
