#!/bin/bash
# instrumented build of the strip kernel only (tuning aid): every other object is the shipped one
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DYMI_STAMPS -x hip -c yolort_amd/csrc/c3_tile.hip -o tools/_bin/c3_tile.stamps.o
objs=$(ls yolort_amd/lib/*.o | grep -v "/c3_tile.o" | grep -v dbg.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libyolort_amd_c3tstamps.so $objs tools/_bin/c3_tile.stamps.o
ls -la tools/_bin/libyolort_amd_c3tstamps.so
