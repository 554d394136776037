#!/bin/bash
# round 4, call z2: the fused stem + body.1 kernel with the planar rows interleaved into super-pixels ONCE per tile (stage 0) instead of once per tap use:
# the bit-identity tests, then a same-box A/B on C2 against the previous form (tools/_ab/libyolort_amd_stemold.so = HEAD~ of csrc/stem_body1_fused.hip)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04z2
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_c3_fused_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "stem or fused or every_conv_launch" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests.txt
run() { cfg=$1; lbl=$2; shift; shift
  env "$@" timeout 600 python bench.py --config $cfg --no-cpu-baseline --per-op $O/perop_${cfg}_$lbl.json 2>$O/err_${cfg}_$lbl.txt | grep '^{"metric' > $O/line_${cfg}_$lbl.json
  python - <<PY
import json
try:
    d = json.loads(open('$O/line_${cfg}_$lbl.json').readline()); r = d['roofline']
    ops = json.load(open('$O/perop_${cfg}_$lbl.json'))
    print('$cfg', '$lbl', 'img/s', d['value'], 'ms/step', d['ms_per_step'], 'serial conv ms', r['serial']['conv_ms_per_step'], 'frac', r['frac'], 'spread %', d['repeats']['spread_pct'], 'first op us (plan.profile)', [(o['name'], round(o['ms'] * 1e3, 1)) for o in ops[:1]])
except Exception as e:
    print('$cfg', '$lbl', 'FAILED', e, open('$O/err_${cfg}_$lbl.txt').read()[-600:])
PY
}
for rep in 1 2 3; do
  run c2 old YOLORT_AMD_LIB=$PWD/tools/_ab/libyolort_amd_stemold.so | tee -a $O/ab.txt
  run c2 new A=1 | tee -a $O/ab.txt
done
