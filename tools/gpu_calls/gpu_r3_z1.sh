#!/bin/bash
# round 3, call z1: 8-wave halo kernel, 16x16 patches (one block per CU with the padded pitch at cin >= 64) against 12x20 (two blocks per CU), same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03y
for rep in 1 2; do
for patch in default 12,20 14,16; do
for cfg in c2 c5; do
if [ $patch = default ]; then unset YOLORT_AMD_H8_PATCH; else export YOLORT_AMD_H8_PATCH=$patch; fi
timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$cfg patch $patch', d['value'], d['ms_per_step'], 'conv serial', r['serial']['conv_ms_per_step'], 'frac', r['frac'])" | tee -a gpurun_out/r03y/patch_ab.txt
done; done; done
