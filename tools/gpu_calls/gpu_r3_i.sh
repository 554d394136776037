#!/bin/bash
# round 3, call i: yolov5m (C3): hidden width 48 padded to 64 -- per-launch parity on the bs-64 plan, A/B of the C3 bench, re-tune of the new shape keys
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -q -s --timeout 800 -p no:cacheprovider -k "(every_conv_launch and m_r60) or c3 or yolov5m" > $O/pytest.log 2>&1
rc=$?; echo "tests rc $rc"; grep -v "^$" $O/pytest.log | tail -12 | cut -c1-250
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'])"
}
run "hidden 48 as is" c3 YOLORT_AMD_PAD_HIDDEN=0
run "hidden 48 -> 64" c3 YOLORT_AMD_PAD_HIDDEN=1
timeout 600 python tools/tune_tiles.py --out $O/tiles_c3.json yolov5_darknet_pan_m_r60:bf16:64:1280:dynamic > $O/tune.log 2>&1; tail -2 $O/tune.log | cut -c1-200
run "hidden 48 -> 64, re-tuned" c3 YOLORT_AMD_PAD_HIDDEN=1 YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_c3.json
run "hidden 48 as is" c3 YOLORT_AMD_PAD_HIDDEN=0
YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_c3.json timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 10 --per-op $O/perop_c3.json > $O/bench_perop.log 2>&1
python - <<'P'
import json
for r in json.load(open('gpurun_out/r03i/perop_c3.json'))[:9]:
    print(r['name'], round(r['ms']*1e3,1), 'us', r.get('tile'), r.get('shape'))
P
