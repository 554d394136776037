#!/bin/bash
# round 3, call s: cheaper plan-cache key (weights_signature): host enqueue time, bench, the tests that swap weights under a live model
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_boundary_gpu.py -m gpu -q --timeout 500 -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], 'host enqueue ms', d['config'].get('host_enqueue_ms_per_step_rank0'))"
done
timeout 300 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('c5', d['value'], d['ms_per_step'], 'host enqueue ms', d['config'].get('host_enqueue_ms_per_step_rank0'))"
