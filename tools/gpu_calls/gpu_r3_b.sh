#!/bin/bash
# round 3, call b: the new golden / conditioned-workload parity tests, then the whole GPU suite, then the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout 600 python -m pytest tests/test_golden_gpu.py -m gpu -q -s --timeout 300 -p no:cacheprovider > $O/pytest_golden.log 2>&1
echo "golden rc $?"; grep -v "^$" $O/pytest_golden.log | tail -40 | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider --deselect tests/test_golden_gpu.py > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -8 $O/pytest_all.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
