#!/bin/bash
# round 5, call J: the legacy releases (r4.0: Focus stem; r3.1: BottleneckCSP, Hardswish / LeakyReLU(0.1) epilogues) on the GPU -- per-launch parity, reference-made
# goldens (fp32 mode exact, fp16 tolerance), then the whole suite (the general epilogues of every kernel family were touched)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05j}
mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py -m gpu -x -q -s -p no:cacheprovider -k "r40 or r31" > $O/pytest_legacy.log 2>&1; grep -v "^$" $O/pytest_legacy.log | tail -30 | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log | cut -c1-300
timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-200 $O/bench_c2.json
