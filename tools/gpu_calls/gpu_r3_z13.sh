#!/bin/bash
# round 3, call z13: multi-block selection with the refinement ladder: the config / post-process / e2e tests, then C3 (64 images) one-block against multi-block
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z13
timeout 150 python -m pytest tests/test_configs_gpu.py tests/test_ops_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "config or post or nms or golden or c3 or c5" 2>&1 | tail -3 | tee gpurun_out/r03z13/tests.txt
for maxn in 32 1000; do
YOLORT_AMD_SEL_MULTI_MAXN=$maxn timeout 100 python bench.py --config c3 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; ok=r['other_kernels']; k=[v for n,v in ok.items() if n.startswith('postprocess')][0]; print('c3 multi_max_n=$maxn', d['value'], d['ms_per_step'], 'post ms', k['ms'], 'cands', k['candidates_per_step'], 'unexplained', d.get('parity',{}).get('unexplained'))" | tee -a gpurun_out/r03z13/ab.txt
done
