#!/bin/bash
# round 4, call n: pipeline depth (plan instances per shape) and hipGraph replay on C2 with the re-tuned table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04n
mkdir -p $O
run() { lbl=$1; shift
  env "$@" timeout 300 python bench.py --config c2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['config'].get('host_enqueue_ms_per_step_rank0'), d['repeats']['spread_pct'])"
}
for rep in 1 2; do
for depth in 3 4 5 6 8; do
run "depth=$depth" YOLORT_AMD_PIPELINE=$depth | tee -a $O/depth.txt
done
run "depth=4 graph" YOLORT_AMD_PIPELINE=4 YOLORT_AMD_GRAPH=1 | tee -a $O/depth.txt
run "depth=6 graph" YOLORT_AMD_PIPELINE=6 YOLORT_AMD_GRAPH=1 | tee -a $O/depth.txt
done
