#!/bin/bash
# round 2, GPU call D: re-tune the tile table with the halo8 kernel in the candidate set, parity tests, bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
date
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $O/tests_ops.log 2>&1; tail -4 $O/tests_ops.log
CASES="32,64,64,160,160,1,1,0 32,128,128,80,80,1,1,0 32,256,256,40,40,1,1,0 32,512,512,20,20,1,1,0 32,64,128,160,160,3,2,1 32,128,256,80,80,3,2,1 32,256,512,40,40,3,2,1 32,1024,512,20,20,1,1,0"
TILES=0,61,68,64,111,112,113,115,116 timeout 600 python tools/conv_bench.py $CASES > $O/conv_bench_i8.txt 2>&1
python - <<'PY'
for l in open("'$O'/conv_bench_i8.txt"):
    if '|' not in l: continue
    case=l[:32].strip(); out=[]
    for p in l[32:].split('|'):
        f=p.split()
        if len(f)>=2: out.append(f[0]+f[1])
    print(case, ' '.join(out))
PY
date
timeout 900 python tools/tune_tiles.py --out $O/tiles_gfx950.json > $O/tune.log 2>&1; tail -2 $O/tune.log
[ -s $O/tiles_gfx950.json ] && cp $O/tiles_gfx950.json yolort_amd/data/tiles_gfx950.json
date
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "fp32_parity or spp or halo8" > $O/tests_par.log 2>&1; grep -E "passed|failed|x2:|x32:|unpaired|FAILED" $O/tests_par.log | cut -c1-400
date
for c in c2 c3 c5; do
timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; python -c "
import json,sys
d=json.load(open('$O/bench_$c.json')); r=d['roofline']
print('$c', d['value'], 'img/s', d['ms_per_step'], 'ms/step conv', r['conv_ms_per_step'], 'frac_bound', r['frac_of_per_layer_bound'], 'TF', r['tflops'], {k[:12]:v['ms'] for k,v in r['other_kernels'].items()})
"
done
date
