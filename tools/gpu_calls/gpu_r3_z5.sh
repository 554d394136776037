#!/bin/bash
# round 3, call z5: multi-block score-prefix selection: post-process tests, then the post-process time per config, one-block against multi-block (same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z5
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py tests/test_e2e_gpu.py -x -q -m gpu -k "post or prefix or config or c5 or c3 or e2e" 2>&1 | tail -4 | tee gpurun_out/r03z5/tests.txt
for single in 1 0; do
for cfg in c5 c3 c2; do
if [ $single = 1 ]; then export YOLORT_AMD_SEL_SINGLE=1; else unset YOLORT_AMD_SEL_SINGLE; fi
timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; ok=r['other_kernels']; k=[v for n,v in ok.items() if n.startswith('postprocess')][0]; print('$cfg single_block=$single', d['value'], d['ms_per_step'], 'post ms', k['ms'], 'cands', k['candidates_per_step'])" | tee -a gpurun_out/r03z5/ab.txt
done; done
