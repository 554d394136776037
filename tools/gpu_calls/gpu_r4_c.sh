#!/bin/bash
# round 4, call c: the whole GPU suite after ABI 5 (kernel-written wire slab, 8-word status, one status+count copy), the reference-own-16-bit assertions, the spread golden
# of yolov5s, tile 134 in the plans; then the default bench line in its new form (repeats + median, raw candidate counts, c2dyn img/s, fp32-mode img/s, parity blocks)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -15 $O/pytest_all.log | cut -c1-400
timeout 300 python -m pytest tests/test_golden_gpu.py -m gpu -q -s -p no:cacheprovider -k "spread or conditioned_workload_16bit or photos_16bit" 2>&1 | grep -v "^$" | cut -c1-700 > $O/golden_verbose.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc $?"; tail -3 $O/bench_c2.err; cut -c1-600 $O/bench_c2.json
YOLORT_AMD_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-500 > $O/bench_c2_graph.txt; cut -c1-300 $O/bench_c2_graph.txt
