#!/bin/bash
# round 3, call t: plan instances / batches in flight, now that the host costs 0.43 ms per batch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for p in 4 3 5 6 8 4; do
YOLORT_AMD_PIPELINE=$p timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('pipeline $p: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], 'host', d['config'].get('host_enqueue_ms_per_step_rank0'))"
done
for g in 1; do
YOLORT_AMD_PIPELINE=6 timeout 300 python bench.py --no-cpu-baseline --steps 200 --graph 1 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('pipeline 6 graph: c2', d['value'], d['ms_per_step'], 'host', d['config'].get('host_enqueue_ms_per_step_rank0'))"
done
