#!/bin/bash
# round 4, call t120: conv_igemm8 in 192-cout blocks (tile 120) on yolov5m's 192-cout shapes (bs 64 at 1280^2, bf16) against the 128 / 256-wide tiles; then the re-tune of yolov5m's entries
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04t120
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "192_cout" -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
export DTYPE=bf16
TILES=143,111,115,151,155,66,120 timeout 300 python tools/conv_bench.py "64,96,192,320,320,3,2,1" "64,192,192,160,160,3,2,1" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
TILES=68,115,111,21,120 timeout 300 python tools/conv_bench.py "64,192,192,160,160,1,1,0" "64,384,192,160,160,1,1,0" 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench.txt
cp yolort_amd/data/tiles_gfx950.json $O/tiles_retuned.json
timeout 900 python tools/tune_tiles.py --out $O/tiles_retuned.json --merge yolov5_darknet_pan_m_r60:bf16:64:1280:dynamic 2>&1 | grep -v amdgpu.ids | tail -2
python - <<'PY'
import json
a=json.load(open('yolort_amd/data/tiles_gfx950.json'))['tiles']; b=json.load(open('gpurun_out/r04t120/tiles_retuned.json'))['tiles']
for k in b:
    if b[k] == 120: print(a.get(k), '-> 120', k)
PY
