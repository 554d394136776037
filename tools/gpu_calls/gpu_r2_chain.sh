#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['launches_per_step'])"
}
for rep in 1 2; do
run "default" A=1
run "chain cv3" YOLORT_AMD_CHAIN_CV3=1
run "no chain" YOLORT_AMD_CHAIN=0
done
