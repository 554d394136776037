#!/bin/bash
# round 4, call i: tile 134 with the explicitly scheduled fragment loop at PF = 0 / 3 / 4 (p_rc only), whole kernel and compute-only (+0x800+0x200)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04i
mkdir -p $O
for pf in 0 3 4; do
echo "PF=$pf" | tee -a $O/conv_bench_pf.txt
YOLORT_AMD_RW2_PF=$pf timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rw2" -p no:cacheprovider 2>&1 | tail -1 | tee -a $O/conv_bench_pf.txt
YOLORT_AMD_RW2_PF=$pf TILES=134,2694,2182 timeout 300 python tools/conv_bench.py 32,64,128,160,160,3,2,1 8,64,128,640,640,3,2,1 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench_pf.txt
done
