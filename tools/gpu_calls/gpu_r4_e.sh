#!/bin/bash
# round 4, call e: tile 136 = tile 134 with a dedicated DMA wave, three patch buffers and compute waves that never wait for memory (one 5-wave block per CU): op tests,
# bit-identity, layer timings against tile 134, same-box A/B on C2 / C5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04e
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rw2" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests_rw2.txt
for rep in 1 2; do
TILES=143,134,136 timeout 300 python tools/conv_bench.py 32,64,128,160,160,3,2,1 8,64,128,640,640,3,2,1 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench_136.txt
done
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['config'].get('host_enqueue_ms_per_step_rank0'), d['repeats']['spread_pct'])"
}
for rep in 1 2; do
run "RW2=1" c2 YOLORT_AMD_RW2=1 | tee -a $O/ab_136.txt
run "RW2=2" c2 YOLORT_AMD_RW2=2 | tee -a $O/ab_136.txt
done
run "RW2=1" c5 YOLORT_AMD_RW2=1 | tee -a $O/ab_136.txt
run "RW2=2" c5 YOLORT_AMD_RW2=2 | tee -a $O/ab_136.txt
