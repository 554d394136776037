#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do for p in 4 8 12 16; do
YOLORT_AMD_PIPELINE=$p timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('pipeline $p: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done; done
for p in 4 8; do
YOLORT_AMD_GRAPH=1 YOLORT_AMD_PIPELINE=$p timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('graph pipeline $p: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
for c in c3 c5; do for p in 4 8; do
YOLORT_AMD_PIPELINE=$p timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('pipeline $p: $c', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done; done
