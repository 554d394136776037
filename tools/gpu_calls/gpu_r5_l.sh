#!/bin/bash
# round 5, call L: the legacy activations as a launch of their own (ymi_act) -- the r6.0 bench back at its level?  legacy per-launch parity / goldens / ops test, then the whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05l}
mkdir -p $O
timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-200 $O/bench_c2.json
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_ops_gpu.py -m gpu -x -q -s -p no:cacheprovider -k "r40 or r31 or legacy" > $O/pytest_legacy.log 2>&1; grep -v "^$" $O/pytest_legacy.log | tail -25 | cut -c1-330
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log | cut -c1-300
