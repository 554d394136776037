#!/bin/bash
# round 4, call h96b: C3 with the eight tile-96 entries in the table against the table without them (same box); per-launch parity of the yolov5m plan and the golden tests with the new table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04h96b
mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "m_r60 or every_conv_launch or cond_m or spread or _m or at_spec" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests.txt
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['repeats']['spread_pct'])"
}
for rep in 1 2; do
run before c3 YOLORT_AMD_TILE_TABLE_PATH=$PWD/tools/_ab/tiles_before96.json | tee -a $O/ab.txt
run tile96 c3 A=1 | tee -a $O/ab.txt
done
