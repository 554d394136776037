#!/bin/bash
# round 3, call v: chain-next on by default: tune the chained shape keys of all four configs (merged into the table afterwards), per-launch parity on every plan
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03v
mkdir -p $O
cp yolort_amd/data/tiles_gfx950.json $O/tiles_merged.json
timeout 1200 python tools/tune_tiles.py --merge --out $O/tiles_merged.json > $O/tune.log 2>&1; tail -2 $O/tune.log | cut -c1-200
YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_merged.json timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -s --timeout 800 -p no:cacheprovider -k "every_conv_launch" 2>&1 | grep -v "^$" | tail -4 | cut -c1-200
for cfg in c2 c3 c5; do
YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_merged.json timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], r['launches_per_step'])"
done
