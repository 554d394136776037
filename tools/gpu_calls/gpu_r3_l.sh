#!/bin/bash
# round 3, call l: the whole GPU suite + smoke at HEAD
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03l
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -12 $O/pytest_all.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
run() { lbl=$1; cfg=$2; shift; shift
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: $cfg', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['config'].get('host_enqueue_ms_per_step_rank0'))"
}
run "SPP G=1" c3 YOLORT_AMD_SPP_G=1
run "SPP default" c3 A=1
