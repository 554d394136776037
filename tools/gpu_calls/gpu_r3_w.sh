#!/bin/bash
# round 3, call w: chain-next A/B per config with the merged table (same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=$PWD/gpurun_out/r03v/tiles_merged.json
for cfg in c5 c2 c5 c2; do
for k in 0 1; do
YOLORT_AMD_CHAIN_NEXT=$k YOLORT_AMD_TILE_TABLE_PATH=$T timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$cfg chain_next=$k', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], r['launches_per_step'])"
done; done
