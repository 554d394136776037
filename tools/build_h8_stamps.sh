#!/bin/bash
# instrumented build of the 8-wave halo kernel only (tuning aid): every other object is the shipped one
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DYMI_STAMPS -x hip -c yolort_amd/csrc/conv_halo8.hip -o tools/_bin/conv_halo8.stamps.o
objs=$(ls yolort_amd/lib/*.o | grep -v conv_halo8.o | grep -v dbg.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libyolort_amd_h8stamps.so $objs tools/_bin/conv_halo8.stamps.o
ls -la tools/_bin/libyolort_amd_h8stamps.so
