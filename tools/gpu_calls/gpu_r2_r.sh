#!/bin/bash
# round 2, GPU call R: re-tune the tile table with the round-2 kernels, A/B old table vs new (same box), graph replay on/off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02r
mkdir -p $O
cp yolort_amd/data/tiles_gfx950.json $O/tiles_old.json
date
timeout 1200 python tools/tune_tiles.py --out $O/tiles_new.json > $O/tune.log 2>&1; tail -2 $O/tune.log | cut -c1-200
date
for t in old new old new; do
YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_$t.json timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('table $t: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
for gr in 0 1; do
YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_new.json timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 --graph $gr 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('graph $gr: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
for c in c3 c5; do for t in old new; do
YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_$t.json timeout 400 python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('table $t: $c', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done; done
date
