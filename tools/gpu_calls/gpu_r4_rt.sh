#!/bin/bash
# round 4, call rt: re-tune of yolov5m's table entries (bf16, bs 64, 1280^2 dynamic) after the epilogue change (cout 96 / 192 / 48 wave tiles are lean now: the wide tiles
# may win where narrow ones were chosen), into a separate table; then same-box A/B on C3: committed table vs re-tuned
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04rt
mkdir -p $O
cp yolort_amd/data/tiles_gfx950.json $O/tiles_retuned.json
date
timeout 900 python tools/tune_tiles.py --out $O/tiles_retuned.json --merge yolov5_darknet_pan_m_r60:bf16:64:1280:dynamic 2>&1 | grep -v amdgpu.ids | tail -4
date
python - <<'PY'
import json
a=json.load(open('yolort_amd/data/tiles_gfx950.json'))['tiles']; b=json.load(open('gpurun_out/r04rt/tiles_retuned.json'))['tiles']
ch=[(k,a.get(k),b[k]) for k in b if a.get(k)!=b[k]]
print(len(ch),'entries differ')
for k,x,y in ch[:80]: print(x,'->',y,k)
PY
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['repeats']['spread_pct'])"
}
for rep in 1 2; do
run committed c3 A=1 | tee -a $O/ab_retune.txt
run retuned c3 YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_retuned.json | tee -a $O/ab_retune.txt
done
