#!/bin/bash
# round 3, call j: host-side cost per batch (enqueue / collect) and the default bench line after the plan-key fix
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
timeout 120 python tools/host_overhead.py 2>/dev/null | tail -1
YOLORT_AMD_GRAPH=1 timeout 120 python tools/host_overhead.py 2>/dev/null | tail -1
for g in 0 1; do
timeout 300 python bench.py --no-cpu-baseline --graph $g 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('graph $g: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], r['shader_clock_mhz_measured'])"
done
timeout 300 python bench.py > $O/bench_c2.json 2>/dev/null; cut -c1-200 $O/bench_c2.json
