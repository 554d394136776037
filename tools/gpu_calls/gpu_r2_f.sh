#!/bin/bash
# round 2, GPU call F: streaming 1x1 kernel -- correctness, micro-benchmarks, re-tune, parity, bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
fmt() { python - "$1" <<'PY'
import sys
for l in open(sys.argv[1]):
    if '|' not in l: continue
    case=l[:32].strip(); out=[]
    for p in l[32:].split('|'):
        f=p.split()
        if len(f)>=5: out.append(f[0]+f[1]+'('+f[4]+'GB/s)')
        elif len(f)>=2: out.append(f[0]+f[1])
    print(case, ' '.join(out))
PY
}
date
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "stream or chained_1x1 or letterbox" > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
CASES="32,64,64,160,160,1,1,0 32,64,64,80,80,1,1,0 32,128,128,80,80,1,1,0 32,128,128,40,40,1,1,0 64,96,96,320,320,1,1,0 64,96,192,160,160,1,1,0"
TILES=0,26,27,12,68,121,122,123,124 timeout 600 python tools/conv_bench.py $CASES > $O/cb_stream.txt 2>&1; fmt $O/cb_stream.txt
date
timeout 900 python tools/tune_tiles.py --out $O/tiles_gfx950.json > $O/tune.log 2>&1; tail -2 $O/tune.log | cut -c1-200
[ -s $O/tiles_gfx950.json ] && cp $O/tiles_gfx950.json yolort_amd/data/tiles_gfx950.json
date
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider > $O/tests_par.log 2>&1; grep -E "passed|failed|x2:|x32:|FAILED|layer outputs|e-0[0-9]  model" $O/tests_par.log | cut -c1-330
for c in c2 c3 c5; do
timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; python -c "
import json,sys
d=json.load(open('$O/bench_$c.json')); r=d['roofline']
print('$c', d['value'], 'img/s', d['ms_per_step'], 'ms/step conv', r['conv_ms_per_step'], 'frac_bound', r['frac_of_per_layer_bound'], 'TF', r['tflops'], {k[:12]:v['ms'] for k,v in r['other_kernels'].items()})
"
done
date
