#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
}
for rep in 1 2; do
run "default" A=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "AMD_OPT_FLUSH=0" AMD_OPT_FLUSH=0
run "HSA_FORCE_FINE_GRAIN_PCIE=1" HSA_FORCE_FINE_GRAIN_PCIE=1
done
