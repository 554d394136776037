#!/bin/bash
# round 5, call A: the fp32 mode's pipelined kernel family (csrc/conv_f32_pipe.hip) -- op tests, fp32 goldens, per-launch tables (new family / register-staged
# kernel), every tile timed on every conv of the C2 plan, the fp32 bench line, and the fp16 headline at the start of the round
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05a
mkdir -p $O
date
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "f32" -p no:cacheprovider > $O/pytest_f32_ops.log 2>&1; echo "f32 ops rc $?"; tail -5 $O/pytest_f32_ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_parity_gpu.py -m gpu -q -s -k "fp32" -p no:cacheprovider > $O/pytest_f32_golden.log 2>&1; echo "f32 golden rc $?"; grep -v "^$" $O/pytest_f32_golden.log | tail -40 | cut -c1-400
date
timeout 600 python tools/f32_layer_profile.py --config c2 --tune --json $O/f32_c2_new.json > $O/f32_layers_c2_tune.csv 2> $O/f32_layers_c2_tune.err; tail -80 $O/f32_layers_c2_tune.csv | cut -c1-400
date
timeout 300 python tools/f32_layer_profile.py --config c2 --json $O/f32_c2_rule.json > $O/f32_layers_c2.csv 2> $O/f32_layers_c2.err; grep "^#" $O/f32_layers_c2.csv
YOLORT_AMD_F32_V1=1 timeout 300 python tools/f32_layer_profile.py --config c2 --json $O/f32_c2_v1.json > $O/f32_layers_c2_v1.csv 2> $O/f32_layers_c2_v1.err; grep "^#" $O/f32_layers_c2_v1.csv
date
timeout 600 python bench.py --config c2 --dtype fp32 > $O/bench_c2_fp32.log 2>&1; grep '^{"metric' $O/bench_c2_fp32.log | tail -1 > $O/bench_c2_fp32.json; cut -c1-1500 $O/bench_c2_fp32.json; tail -3 $O/bench_c2_fp32.log | cut -c1-300
date
timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2_fp16.log 2>&1; grep '^{"metric' $O/bench_c2_fp16.log | tail -1 > $O/bench_c2_fp16.json; cut -c1-400 $O/bench_c2_fp16.json
date
