#!/bin/bash
# round 3, call e: first GPU run of the fused stem + body.1 launch: its tests, then same-box A/B of the C2 bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 900 python -m pytest tests/test_c3_fused_gpu.py tests/test_parity_gpu.py tests/test_e2e_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "fused_stem or (every_conv_launch and s_r60) or planar" > $O/pytest.log 2>&1
rc=$?; echo "tests rc $rc"; grep -v "^$" $O/pytest.log | tail -25 | cut -c1-300
[ $rc -ne 0 ] && exit 0
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], r['launches_per_step'])"
}
for rep in 1 2 3; do
run "separate stem / body.1" YOLORT_AMD_FUSE_STEM=0
run "fused stem + body.1" YOLORT_AMD_FUSE_STEM=1
done
YOLORT_AMD_FUSE_STEM=1 timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 50 --per-op $O/perop.json > $O/bench_perop.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_c2.json > /tmp/ps_c2.log 2>&1)
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python tools/rocprof_summary.py $db > $O/rocprof_summary_c2.csv 2>> $O/err.log
python tools/layer_table.py --ops $O/ops_c2.json --stats $db > $O/layer_table_c2.csv 2>> $O/err.log
head -12 $O/layer_table_c2.csv | cut -c1-220; tail -16 $O/layer_table_c2.csv | cut -c1-200; tail -3 $O/err.log
