#!/bin/bash
# instrumented build of the resident-weights 3x3 kernel only (tuning aid): every other object is the shipped one
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -DYMI_STAMPS -x hip -c yolort_amd/csrc/conv3x3_res.hip -o tools/_bin/conv3x3_res.stamps.o
objs=$(ls yolort_amd/lib/*.o | grep -v conv3x3_res.o | grep -v dbg.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libyolort_amd_r3stamps.so $objs tools/_bin/conv3x3_res.stamps.o
ls -la tools/_bin/libyolort_amd_r3stamps.so
