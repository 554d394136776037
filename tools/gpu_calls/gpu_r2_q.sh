#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
}
for rep in 1 2; do
for q in 2 4 8 16; do for p in 4 8; do run "hwq $q pipeline $p" GPU_MAX_HW_QUEUES=$q YOLORT_AMD_PIPELINE=$p; done; done
run "post hi, pipeline 8" YOLORT_AMD_POST_PRIORITY=-1 YOLORT_AMD_PIPELINE=8
run "post hi, pipeline 4" YOLORT_AMD_POST_PRIORITY=-1 YOLORT_AMD_PIPELINE=4
run "conv hi, pipeline 8" YOLORT_AMD_CONV_PRIORITY=-1 YOLORT_AMD_PIPELINE=8
done
