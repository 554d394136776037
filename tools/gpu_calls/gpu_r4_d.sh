#!/bin/bash
# round 4, call d: first GPU run of the K-split stride-2 register-weights 3x3 (tile 135): op tests, layer timings against the table's tiles, same-box A/B of C2 with
# YOLORT_AMD_RW3=0/1; the four tests call c failed (criteria / test fixed); C3 and C5 bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rw3 or rw2" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests_rw3.txt
timeout 600 python -m pytest tests/test_boundary_gpu.py tests/test_golden_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300 | tee $O/tests_fixed.txt
TILES=142,155,111,135 timeout 300 python tools/conv_bench.py 32,128,128,80,80,3,2,1 32,128,256,80,80,3,2,1 8,128,256,320,320,3,2,1 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_rw3.txt
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['config'].get('host_enqueue_ms_per_step_rank0'), d['repeats']['spread_pct'])"
}
for rep in 1 2; do
run "RW3=0" c2 YOLORT_AMD_RW3=0 | tee -a $O/ab_rw3.txt
run "RW3=1" c2 YOLORT_AMD_RW3=1 | tee -a $O/ab_rw3.txt
done
run "RW3=1 GRAPH=1" c2 YOLORT_AMD_RW3=1 YOLORT_AMD_GRAPH=1 | tee -a $O/ab_rw3.txt
run "RW3=0" c5 YOLORT_AMD_RW3=0 | tee -a $O/ab_rw3.txt
run "RW3=1" c5 YOLORT_AMD_RW3=1 | tee -a $O/ab_rw3.txt
run "RW3=0" c3 YOLORT_AMD_RW3=0 | tee -a $O/ab_rw3.txt
