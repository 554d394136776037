#!/bin/bash
# round 5, call I: the C pass over the image list (sig_ext.cpp images / record_stream) on top of the two-call C submit: e2e / boundary / config / golden / dist-free suites, host cost per batch
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05i}
mkdir -p $O
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_boundary_gpu.py tests/test_configs_gpu.py tests/test_golden_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest_e2e.log 2>&1; tail -6 $O/pytest_e2e.log | cut -c1-300
timeout 300 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; tail -5 $O/host_overhead.txt
YOLORT_AMD_SIG_EXT=0 YOLORT_AMD_C_SUBMIT=0 timeout 300 python tools/host_overhead.py > $O/host_overhead_python.txt 2>&1; tail -4 $O/host_overhead_python.txt
