#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02o
mkdir -p $O
for fl in 0 81; do
echo "== LDS floor $fl KB"
YOLORT_AMD_LDS_FLOOR_KB=$fl TILES=93,91,92,35 timeout 200 python tools/conv_bench.py 32,128,128,40,40,3,1,1 32,64,64,80,80,3,1,1 32,256,256,20,20,3,1,1 2>&1 | grep -v amdgpu.ids
YOLORT_AMD_LDS_FLOOR_KB=$fl TILES=21,24,68,65 timeout 200 python tools/conv_bench.py 32,128,128,40,40,1,1,0 32,256,256,40,40,1,1,0 32,128,128,80,80,1,1,0 2>&1 | grep -v amdgpu.ids
done
python - <<'PY'
import ctypes, torch
# occupancy of a few kernels as the runtime sees it
hip = ctypes.CDLL("libamdhip64.so")
print(torch.cuda.get_device_properties(0))
PY
