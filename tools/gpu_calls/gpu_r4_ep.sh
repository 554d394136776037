#!/bin/bash
# round 4, call ep: the lean epilogue for wave tiles whose trailing sub-tiles lie past cout (cout = 96 / 192: yolov5m) + hardware pair conversion in the general epilogue.
# Conv tests + per-launch parity (every conv launch of the plans vs the oracle's layers), then same-box A/B on C3 (and C2 / C5 as controls) against HEAD~'s library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04ep2
mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py tests/test_c3_fused_gpu.py -x -q -m gpu -k "conv or every_conv_launch or fused or chain" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests.txt
run() { cfg=$1; lbl=$2; shift; shift
  env "$@" timeout 600 python bench.py --config $cfg --no-cpu-baseline 2>$O/err_${cfg}_$lbl.txt | grep '^{"metric' > $O/line_${cfg}_$lbl.json
  python - <<PY
import json
try:
    d = json.loads(open('$O/line_${cfg}_$lbl.json').readline()); r = d['roofline']
    print('$cfg', '$lbl', 'img/s', d['value'], 'ms/step', d['ms_per_step'], 'serial conv ms', r['serial']['conv_ms_per_step'], 'frac', r['frac'], 'spread %', d['repeats']['spread_pct'])
except Exception as e:
    print('$cfg', '$lbl', 'FAILED', e, open('$O/err_${cfg}_$lbl.txt').read()[-600:])
PY
}
for rep in 1 2; do
  run c3 old YOLORT_AMD_LIB=$PWD/tools/_ab/libyolort_amd_old.so | tee -a $O/ab.txt
  run c3 new A=1 | tee -a $O/ab.txt
done
run c2 old YOLORT_AMD_LIB=$PWD/tools/_ab/libyolort_amd_old.so | tee -a $O/ab.txt
run c2 new A=1 | tee -a $O/ab.txt


