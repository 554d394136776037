#!/bin/bash
# round 4, call u3: the tiled letterbox kernels with hardware pair conversions for the 16-bit output (bf16: v_cvt_pk_bf16_f32 instead of ~7 VALU of software rounding per value):
# bit-identity against the per-pixel kernel for both output types, then the kernel alone on the C3 batch (bf16) and the C2-like dynamic batch (fp16)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04u4
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "letterbox" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests_letterbox.txt
for w in c3 c2dyn; do
  for rep in 1 2; do
    for k in default d4; do
      if [ $k = default ]; then env -u YOLORT_AMD_LETTERBOX timeout 120 python tools/letterbox_bench.py $w 40 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_$w.txt
      else YOLORT_AMD_LETTERBOX=$k timeout 120 python tools/letterbox_bench.py $w 40 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_$w.txt; fi
    done
  done
done
