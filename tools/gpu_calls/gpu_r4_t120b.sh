#!/bin/bash
# round 4, call t120b: C3 with the five tile-120 entries in the table against the table without them (same box); per-launch parity of the yolov5m plan
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04t120b
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "every_conv_launch and m_r60" -p no:cacheprovider 2>&1 | tail -2 | cut -c1-300 | tee $O/tests.txt
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['repeats']['spread_pct'])"
}
for rep in 1 2; do
run before c3 YOLORT_AMD_TILE_TABLE_PATH=$PWD/tools/_ab/tiles_before120.json | tee -a $O/ab.txt
run tile120 c3 A=1 | tee -a $O/ab.txt
done
