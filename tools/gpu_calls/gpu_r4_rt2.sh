#!/bin/bash
# round 4, call rt2: re-tune of yolov5m's entries with tile 96 among the candidates, into a separate table (only its tile-96 wins are merged into the committed table afterwards)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04rt2
mkdir -p $O
cp yolort_amd/data/tiles_gfx950.json $O/tiles_retuned.json
timeout 900 python tools/tune_tiles.py --out $O/tiles_retuned.json --merge yolov5_darknet_pan_m_r60:bf16:64:1280:dynamic 2>&1 | grep -v amdgpu.ids | tail -3
python - <<'PY'
import json
a=json.load(open('yolort_amd/data/tiles_gfx950.json'))['tiles']; b=json.load(open('gpurun_out/r04rt2/tiles_retuned.json'))['tiles']
for k in b:
    if b[k] == 96: print(a.get(k), '-> 96', k)
PY
