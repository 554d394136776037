#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02n
mkdir -p $O
for c in 32,128,128,40,40,93 32,64,64,80,80,93 32,256,256,20,20,92 32,32,32,160,160,94; do
YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_h8stamps.so timeout 300 python tools/stamp_h8.py $c >> $O/stamps.txt 2>&1
done
grep -v amdgpu.ids $O/stamps.txt
