#!/bin/bash
# round 3, call z12: multi-block selection with the refinement ladder: post-process tests, then C5 post-process time one-block / multi-block (same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z12
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "post or prefix or c5" 2>&1 | tail -3 | tee gpurun_out/r03z12/tests.txt
for single in 1 0; do
if [ $single = 1 ]; then export YOLORT_AMD_SEL_SINGLE=1; else unset YOLORT_AMD_SEL_SINGLE; fi
timeout 200 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; ok=r['other_kernels']; k=[v for n,v in ok.items() if n.startswith('postprocess')][0]; print('c5 single_block=$single', d['value'], d['ms_per_step'], 'post ms', k['ms'], 'cands', k['candidates_per_step'], 'unexplained', d.get('parity',{}).get('unexplained'))" | tee -a gpurun_out/r03z12/ab.txt
done
