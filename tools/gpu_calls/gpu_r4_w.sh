#!/bin/bash
# round 4, call w: SPP pool pyramid with 512 / 1024-thread blocks (was 256): exactness tests, then same-box A/B per config (YOLORT_AMD_SPP_NT=256 = the old block size)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04w
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "spp or pool or at_spec" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests_spp.txt
run() { cfg=$1; lbl=$2; shift; shift
  env "$@" timeout 600 python bench.py --config $cfg --no-cpu-baseline --per-op $O/perop_${cfg}_$lbl.json 2>$O/err_${cfg}_$lbl.txt | grep '^{"metric' > $O/line_${cfg}_$lbl.json
  python - <<PY
import json
try:
    d = json.loads(open('$O/line_${cfg}_$lbl.json').readline()); r = d['roofline']
    ops = json.load(open('$O/perop_${cfg}_$lbl.json'))
    pool = [round(o['ms'] * 1e3, 1) for o in ops if 'pool' in o['name']]
    print('$cfg', '$lbl', 'img/s', d['value'], 'ms/step', d['ms_per_step'], 'serial conv ms', r['serial']['conv_ms_per_step'], 'frac', r['frac'], 'spread %', d['repeats']['spread_pct'], 'pool us (plan.profile)', pool)
except Exception as e:
    print('$cfg', '$lbl', 'FAILED', e, open('$O/err_${cfg}_$lbl.txt').read()[-600:])
PY
}
for rep in 1 2; do
  run c2 nt256 YOLORT_AMD_SPP_NT=256 | tee -a $O/ab.txt
  run c2 new A=1 | tee -a $O/ab.txt
done
run c5 nt256 YOLORT_AMD_SPP_NT=256 | tee -a $O/ab.txt
run c5 new A=1 | tee -a $O/ab.txt
run c3 nt256 YOLORT_AMD_SPP_NT=256 | tee -a $O/ab.txt
run c3 new A=1 | tee -a $O/ab.txt
run c3 nt512 YOLORT_AMD_SPP_NT=512 | tee -a $O/ab.txt
