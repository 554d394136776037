#!/bin/bash
# round 2, GPU call B: fixed tests, halo8 correctness + micro-benchmarks + timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
date
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $O/tests_ops.log 2>&1; tail -5 $O/tests_ops.log
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_boundary_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider > $O/tests_par.log 2>&1; grep -E "passed|failed|x2:|x32:|bs32 fp16|FAILED" $O/tests_par.log | cut -c1-300
date
CASES="32,32,32,160,160,3,1,1 32,64,64,80,80,3,1,1 32,128,128,40,40,3,1,1 32,256,256,20,20,3,1,1"
TILES=36,35,34,64,61,91,92,93,94,95 timeout 600 python tools/conv_bench.py $CASES > $O/conv_bench_h8.txt 2>&1; cat $O/conv_bench_h8.txt | cut -c1-900
date
YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_h8stamps.so timeout 300 python tools/stamp_h8.py 32,128,128,40,40,91 32,128,128,40,40,92 32,64,64,80,80,92 32,256,256,20,20,91 32,256,256,20,20,92 > $O/stamps_h8.txt 2>&1; cat $O/stamps_h8.txt
date
