#!/bin/bash
# round 3, call p: the default bench line at HEAD (with profiles/conv_traffic.json of the r03z PMC passes) + the golden tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
timeout 600 python -m pytest tests/test_golden_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -3
