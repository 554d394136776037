#!/bin/bash
# round 5, call B: the pipelined fp32 tiles with the lane swap (bit-identical to the register-staged kernel): op tests, every fp32 golden, per-launch table with the
# shape rule, depth sweep, rocprofv3 kernel statistics + SQ counters of the fp32 plan (matrix-pipe busy cycles, GPU-active cycles -> effective clock)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05b
mkdir -p $O
date
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "f32" -p no:cacheprovider > $O/pytest_f32_ops.log 2>&1; echo "f32 ops rc $?"; tail -3 $O/pytest_f32_ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_parity_gpu.py -m gpu -q -s -k "fp32" -p no:cacheprovider > $O/pytest_f32_golden.log 2>&1; echo "f32 golden rc $?"; grep -a "fp32 parity mode:\|passed\|failed" $O/pytest_f32_golden.log | cut -c1-330 | tail -24
date
for d in 2 4; do timeout 300 python tools/f32_layer_profile.py --config c2 --depth $d --steps 24 > $O/f32_layers_c2_d$d.csv 2> $O/f32_layers_c2_d$d.err; grep "^#" $O/f32_layers_c2_d$d.csv; done
timeout 300 python tools/f32_layer_profile.py --config c2 --depth 3 --steps 24 --json $O/f32_c2.json > $O/f32_layers_c2.csv 2> $O/f32_layers_c2.err; grep "^#" $O/f32_layers_c2.csv
date
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o r -- python $GRAFT_REPO_ROOT/tools/f32_layer_profile.py --config c2 --depth 1 --steps 4 > /tmp/ps_f32.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmc_f32 -o r -- python $GRAFT_REPO_ROOT/tools/f32_layer_profile.py --config c2 --depth 1 --steps 4 > /tmp/pmc_f32.log 2>&1) || echo "pmc pass failed"
python tools/rocprof_summary.py $(find /tmp/prof_f32 -name "*.db" | head -1) --pmc $(find /tmp/pmc_f32 -name "*.db" | head -1) > $O/rocprof_summary_c2_fp32.csv 2> $O/rocprof_err.log; head -12 $O/rocprof_summary_c2_fp32.csv | cut -c1-200; grep "conv_f32_pipe" $O/rocprof_summary_c2_fp32.csv | grep -v "^\"void" | head; tail -3 $O/rocprof_err.log
date
timeout 600 python bench.py --config c2 --dtype fp32 > $O/bench_c2_fp32.log 2>&1; grep '^{"metric' $O/bench_c2_fp32.log | tail -1 > $O/bench_c2_fp32.json; cut -c1-700 $O/bench_c2_fp32.json
date
