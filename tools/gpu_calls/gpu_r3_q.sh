#!/bin/bash
# round 3, call q: output stores of the lean epilogues nontemporal (tools/_ab/lib_nt.so, -DYMI_NT_STORES) vs plain, same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'])"
}
for rep in 1 2 3; do
run "plain stores" A=1
run "nontemporal stores" YOLORT_AMD_LIB=$PWD/tools/_ab/lib_nt.so
done
YOLORT_AMD_LIB=$PWD/tools/_ab/lib_nt.so timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 500 -p no:cacheprovider -k "every_conv_launch and s_r60" 2>&1 | tail -2
