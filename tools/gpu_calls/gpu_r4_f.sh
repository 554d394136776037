#!/bin/bash
# round 4, call f: explicitly scheduled LDS fragment reads (inline-asm ds_read_b128, counted lgkmcnt) in tiles 134 / 136: prefetch depth 2 / 3 / 4 units
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04f
mkdir -p $O
for pf in 2 3 4; do
YOLORT_AMD_RW2_PF=$pf timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rw2" -p no:cacheprovider 2>&1 | tail -1 | tee -a $O/tests_rw2.txt
echo "PF=$pf" | tee -a $O/conv_bench_pf.txt
YOLORT_AMD_RW2_PF=$pf TILES=143,134,136 timeout 300 python tools/conv_bench.py 32,64,128,160,160,3,2,1 8,64,128,640,640,3,2,1 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench_pf.txt
done
