#!/bin/bash
# round 3, call r: exact radix refinement of the score-prefix selection: post-process tests, C3 / C5 / C2 bench lines (post-process time alone)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_e2e_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "post or nms or prefix or c5 or c3 or l6 or crowd or topk or capacity" 2>&1 | tail -4
for cfg in c5 c3 c2; do
timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], {k[:11]:v.get('ms') for k,v in r['other_kernels'].items() if isinstance(v,dict) and 'ms' in v}, d['config']['candidates_per_step_rank0'])"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c5 --steps 8 > /tmp/ps_c5.log 2>&1)
python tools/rocprof_summary.py $(find /tmp/prof_c5 -name "*.db" | head -1) 2>/dev/null | grep -i "select\|sort_image\|rank_image\|nms_seg\|scatter\|gather\|find_seg" | cut -c1-160
