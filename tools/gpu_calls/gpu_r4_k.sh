#!/bin/bash
# round 4, call k: first GPU run of the row-streaming 3x3 (conv3x3_rs.hip, tiles 137 / 138): op tests + bit-identity, layer timings against tiles 133 / 134 with counted waits
# and with everything drained per step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04k
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3_rs" -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300 | tee $O/tests_rs.txt
for cnt in 1 0; do
echo "COUNTED=$cnt" | tee -a $O/conv_bench_rs.txt
YOLORT_AMD_RS_COUNTED=$cnt TILES=134,138 timeout 300 python tools/conv_bench.py 32,64,128,160,160,3,2,1 8,64,128,640,640,3,2,1 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench_rs.txt
YOLORT_AMD_RS_COUNTED=$cnt TILES=133,137 timeout 300 python tools/conv_bench.py 32,64,64,80,80,3,1,1 8,64,64,320,320,3,1,1 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench_rs.txt
done
