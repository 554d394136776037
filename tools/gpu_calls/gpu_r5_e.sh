#!/bin/bash
# round 5, call E: the whole GPU suite with hipGraph replay as the default, smoke, the headline line (>= 2 s of timed regions) and the fp32 line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05e
mkdir -p $O
date
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -12 $O/pytest_all.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
date
timeout 600 python bench.py --config c2 > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-900 $O/bench_c2.json; tail -2 $O/bench_c2.log | cut -c1-300
date
timeout 600 python bench.py --config c2 --dtype fp32 > $O/bench_c2_fp32.log 2>&1; grep '^{"metric' $O/bench_c2_fp32.log | tail -1 > $O/bench_c2_fp32.json; cut -c1-600 $O/bench_c2_fp32.json
date
