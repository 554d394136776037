#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02u
mkdir -p $O
cp yolort_amd/data/tiles_gfx950.json $O/tiles_old.json
timeout 1200 python tools/tune_tiles.py --out $O/tiles_new.json > $O/tune.log 2>&1; tail -1 $O/tune.log | cut -c1-200
for t in old new old new; do
YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_$t.json timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('table $t: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
