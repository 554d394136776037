#!/bin/bash
# round 4, call u: the persistent DMA-staged letterbox (letterbox_tile3_kernel; YOLORT_AMD_LETTERBOX=dma8 / dma4): bit-identity tests, then the kernel alone on the C3 batch
# (64 images, 8 cycled shapes -> 1280^2 bf16) and the C2-like dynamic batch, against the shipped tile2 kernel; ablations (no loads / no resampling) of both
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04u${CALL_TAG:-}
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "letterbox" -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300 | tee $O/tests_letterbox.txt
for w in c3 c2dyn; do
  for rep in 1 2; do
    for k in default d8 d4 dma8; do
      if [ $k = default ]; then env -u YOLORT_AMD_LETTERBOX LB_COPY_REF=$([ $rep = 1 ] && echo 1) timeout 120 python tools/letterbox_bench.py $w 40 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_$w.txt
      else YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py $w 40 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_$w.txt; fi
    done
  done
  for d in 1 2 3; do
    for k in 4 d8 d4; do
      YOLORT_AMD_LB_DEBUG=$d YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py $w 40 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_$w.txt
    done
  done
done
