#!/bin/bash
# round 4, call w2: SPP pool at yolov5m's 40 x 40 maps: G = 2 (32-byte runs per pixel, 102 KiB: one block per CU; the default there) against G = 1 (16-byte runs, 51 KiB: three blocks per CU), both with 1024-thread blocks
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04w3
mkdir -p $O
run() { cfg=$1; lbl=$2; shift; shift
  env "$@" timeout 600 python bench.py --config $cfg --no-cpu-baseline --per-op $O/perop_${cfg}_$lbl.json 2>$O/err_${cfg}_$lbl.txt | grep '^{"metric' > $O/line_${cfg}_$lbl.json
  python - <<PY
import json
d = json.loads(open('$O/line_${cfg}_$lbl.json').readline()); r = d['roofline']
ops = json.load(open('$O/perop_${cfg}_$lbl.json'))
print('$cfg', '$lbl', 'img/s', d['value'], 'serial conv ms', r['serial']['conv_ms_per_step'], 'pool us (plan.profile)', [round(o['ms'] * 1e3, 1) for o in ops if 'pool' in o['name']])
PY
}
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "spp" -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
run c3 new A=1 | tee -a $O/ab.txt


