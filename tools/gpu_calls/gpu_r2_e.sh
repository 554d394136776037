#!/bin/bash
# round 2, GPU call E: large-M (C3-size) micro-benchmarks of the 8-wave kernels, re-tune, letterbox test, parity, bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
fmt() { python - "$1" <<'PY'
import sys
for l in open(sys.argv[1]):
    if '|' not in l: continue
    case=l[:32].strip(); out=[]
    for p in l[32:].split('|'):
        f=p.split()
        if len(f)>=3: out.append(f[0]+f[1]+'('+f[2]+'TF)')
    print(case, ' '.join(out))
PY
}
date
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "letterbox or software_pipelined or halo8" > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
CASES="64,96,96,320,320,3,1,1 64,192,192,160,160,3,1,1 64,384,384,80,80,3,1,1"
TILES=0,34,35,61,68,91,92,95 timeout 600 python tools/conv_bench.py $CASES > $O/cb_3x3.txt 2>&1; fmt $O/cb_3x3.txt
CASES="64,192,192,160,160,1,1,0 64,384,384,80,80,1,1,0 64,768,768,40,40,1,1,0 64,96,192,320,320,3,2,1 64,192,384,160,160,3,2,1"
TILES=0,61,66,68,111,112,115,116 timeout 600 python tools/conv_bench.py $CASES > $O/cb_i8.txt 2>&1; fmt $O/cb_i8.txt
date
timeout 900 python tools/tune_tiles.py --out $O/tiles_gfx950.json > $O/tune.log 2>&1; tail -2 $O/tune.log | cut -c1-200
[ -s $O/tiles_gfx950.json ] && cp $O/tiles_gfx950.json yolort_amd/data/tiles_gfx950.json
date
for c in c2 c3 c5; do
timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; python -c "
import json,sys
d=json.load(open('$O/bench_$c.json')); r=d['roofline']
print('$c', d['value'], 'img/s', d['ms_per_step'], 'ms/step conv', r['conv_ms_per_step'], 'frac_bound', r['frac_of_per_layer_bound'], 'TF', r['tflops'], {k[:12]:v['ms'] for k,v in r['other_kernels'].items()})
"
done
date
