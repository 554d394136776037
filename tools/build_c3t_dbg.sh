#!/bin/bash
# tuning builds of the strip kernel with one ingredient removed (C3T_DBG, csrc/c3_tile.hip): every other object is the shipped one
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
objs=$(ls yolort_amd/lib/*.o | grep -v "/c3_tile.o" | grep -v dbg.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DC3T_DBG=$n -x hip -c yolort_amd/csrc/c3_tile.hip -o tools/_bin/c3_tile.dbg$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libyolort_amd_c3tdbg$n.so $objs tools/_bin/c3_tile.dbg$n.o &
done
wait
ls tools/_bin/*.so
