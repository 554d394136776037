#!/bin/bash
# round 4, call b: first GPU run of the stride-2 register-weights 3x3 (tile 134): op tests, bit-identity, same-box A/B of C2 / C5 with YOLORT_AMD_RW2=0/1; GEMM yardstick with
# the launches captured in a graph; host profile of the submit path
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rw2" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests_rw2.txt
timeout 600 python -m pytest tests/test_boundary_gpu.py -x -q -m gpu -k "hook" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests_hook.txt
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['config'].get('host_enqueue_ms_per_step_rank0'))"
}
for rep in 1 2 3; do
run "RW2=0" c2 YOLORT_AMD_RW2=0 | tee -a $O/ab_rw2.txt
run "RW2=1" c2 YOLORT_AMD_RW2=1 | tee -a $O/ab_rw2.txt
done
run "RW2=0" c5 YOLORT_AMD_RW2=0 | tee -a $O/ab_rw2.txt
run "RW2=1" c5 YOLORT_AMD_RW2=1 | tee -a $O/ab_rw2.txt
TILES=143,111,134 timeout 300 python tools/conv_bench.py 32,64,128,160,160,3,2,1 8,64,128,640,640,3,2,1 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench_rw2.txt
for cfg in c2; do
  timeout 300 python tools/gemm_yardstick.py $cfg 2>/dev/null > $O/gemm_yardstick_$cfg.txt; echo "yardstick $cfg rc $?"
done
timeout 300 python tools/host_profile.py 2>/dev/null | head -60 > $O/host_profile.txt
