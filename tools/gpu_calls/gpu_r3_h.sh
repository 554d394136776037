#!/bin/bash
# round 3, call h: same-box A/B of two builds of the fused stem kernel (tools/_ab/lib_base.so = step-by-step stages, lib_pipe.so = loads batched ahead of the MFMAs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for v in base pipe; do
echo -n "$v: "; YOLORT_AMD_LIB=$PWD/tools/_ab/lib_$v.so timeout 120 python tools/stem_bench.py 100 2>/dev/null | tail -1
done; done
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], r['launches_per_step'])"
}
for rep in 1 2; do
for v in base pipe; do run $v YOLORT_AMD_LIB=$PWD/tools/_ab/lib_$v.so; done; done
