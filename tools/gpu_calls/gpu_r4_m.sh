#!/bin/bash
# round 4, call m: re-tune of the four configs with this round's candidates (tiles 134-138, the row-transposed tiles for the upsampled copy) into a SEPARATE table, then same-box
# A/Bs of every config: committed table vs re-tuned table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
cp yolort_amd/data/tiles_gfx950.json $O/tiles_retuned.json
date
timeout 1500 python tools/tune_tiles.py --out $O/tiles_retuned.json --merge 2>&1 | grep -v amdgpu.ids | tail -6
date
python - <<'PY'
import json
a=json.load(open('yolort_amd/data/tiles_gfx950.json'))['tiles']; b=json.load(open('gpurun_out/r04m/tiles_retuned.json'))['tiles']
ch=[(k,a.get(k),b[k]) for k in b if a.get(k)!=b[k]]
print(len(ch),'entries differ'); 
for k,x,y in ch[:80]: print(x,'->',y,k)
PY
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['repeats']['spread_pct'])"
}
for cfg in c2 c5 c3; do
for rep in 1 2; do
run committed $cfg A=1 | tee -a $O/ab_retune.txt
run retuned $cfg YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_retuned.json YOLORT_AMD_RW3=1 | tee -a $O/ab_retune.txt
done; done
