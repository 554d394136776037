#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "spp" 2>&1 | tail -3
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
}
for rep in 1 2 3; do
run "spp G=4 (old)" YOLORT_AMD_SPP_G=4
run "spp G=1" YOLORT_AMD_SPP_G=1
done
for c in c3 c5; do for g in 4 1; do
YOLORT_AMD_SPP_G=$g timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('spp G=$g: $c', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done; done
