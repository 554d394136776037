#!/bin/bash
# round 3, call u: Bottleneck j's 3x3 carrying Bottleneck j+1's 1x1 (8-wave halo kernel with a chained 1x1, YOLORT_AMD_CHAIN_NEXT=1): per-launch parity, tune, A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03u
mkdir -p $O
YOLORT_AMD_CHAIN_NEXT=1 timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -s --timeout 500 -p no:cacheprovider -k "every_conv_launch and (s_r60 or n_r60)" 2>&1 | grep -v "^$" | tail -6 | cut -c1-200
YOLORT_AMD_CHAIN_NEXT=1 timeout 600 python tools/tune_tiles.py --out $O/tiles_chain.json yolov5_darknet_pan_s_r60:fp16:32:640 > $O/tune.log 2>&1; tail -1 $O/tune.log
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], r['launches_per_step'], d['config']['detections_per_step_rank0'])"
}
for rep in 1 2 3; do
run "separate" YOLORT_AMD_CHAIN_NEXT=0
run "chain next" YOLORT_AMD_CHAIN_NEXT=1 YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_chain.json
run "chain next 128 only" YOLORT_AMD_CHAIN_NEXT=128 YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_chain.json
done
python - <<'P'
import json
old=json.load(open('yolort_amd/data/tiles_gfx950.json'))['tiles']; new=json.load(open('gpurun_out/r03u/tiles_chain.json'))
for k,v in new['tiles'].items():
    if 'chain1' in k and 'k3x3' in k: print(k[:70], v, new['us_per_candidate'].get(k))
P
