#!/bin/bash
# round 3, call k: patch-shape sweep of the 8-wave LDS-halo 3x3 kernel on the three yolov5s shapes (blocks per launch vs 512 slots)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
sweep() { case=$1; shift
  for p in "$@"; do echo -n "patch $p: "; YOLORT_AMD_H8_PATCH=$p TILES=93,92 timeout 60 python tools/conv_bench.py $case 2>/dev/null | tail -1; done
}
echo "== 64->64 @80x80 (16x16 = 800 blocks)"; sweep 32,64,64,80,80,3,1,1 "16,16" "10,20" "20,10" "8,20" "10,16" "12,20" "8,32" "5,40"
echo "== 128->128 @40x40 (6x40 = 448 blocks)"; sweep 32,128,128,40,40,3,1,1 "6,40" "5,40" "10,20" "8,20" "7,20" "4,40" "8,40" "14,14"
echo "== 256->256 @20x20 (10x20 = 256 blocks)"; sweep 32,256,256,20,20,3,1,1 "10,20" "5,20" "7,20" "10,10" "4,20" "12,20" "20,10"
