#!/bin/bash
# round 3, call o: SiLU of EVERY epilogue with scalar fp32 instructions (lib_scalar) vs packed (lib_pkdefault; the fused stem's stage 1 scalar in both), same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'])"
}
for rep in 1 2 3; do
for v in pkdefault scalar; do run $v YOLORT_AMD_LIB=$PWD/tools/_ab/lib_$v.so; done; done
for v in pkdefault scalar; do echo -n "$v: "; YOLORT_AMD_LIB=$PWD/tools/_ab/lib_$v.so timeout 120 python tools/stem_bench.py 100 2>/dev/null | tail -1; done
YOLORT_AMD_LIB=$PWD/tools/_ab/lib_scalar.so timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_c3_fused_gpu.py -m gpu -q --timeout 500 -p no:cacheprovider -k "(every_conv_launch and s_r60) or fused" 2>&1 | tail -3
