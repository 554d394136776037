#!/bin/bash
# round 3, call n: the fused stem's stage-1 SiLU with packed fp32 instructions (lib_sbpk1) vs scalar ones (lib_sbpk0), same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in sbpk1 sbpk0; do
echo -n "$v: "; YOLORT_AMD_LIB=$PWD/tools/_ab/lib_$v.so timeout 120 python tools/stem_bench.py 100 2>/dev/null | tail -1
done; done
