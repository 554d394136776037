#!/bin/bash
# round 5, call F: the new tests of the second half (linear-regime goldens s / m, rs counted-wait stress, int64 category ids, fp32 SPP pool, global canvas) + the bench line with graph replay
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05f
mkdir -p $O
date
timeout 900 python -m pytest tests/test_golden_gpu.py -m gpu -q -s -k "linear_regime" -p no:cacheprovider > $O/pytest_lin.log 2>&1; echo "lin rc $?"; grep -a "lin_\|passed\|failed" $O/pytest_lin.log | cut -c1-700 | tail -12
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -m gpu -q -k "row_streaming or category_ids or spp_pool or global_canvas or f32" -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "new rc $?"; tail -6 $O/pytest_new.log | cut -c1-300
date
timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f/bench_c2.json"))
print(d["value"], d["ms_per_step"], d["serial_images_per_s"], d["config"]["host_enqueue_ms_per_step_rank0"], d["config"]["serving_mode_rank0"], d["roofline"]["frac"], d["roofline"]["serial"]["conv_ms_per_step"])
PY
timeout 300 python tools/f32_layer_profile.py --config c2 --depth 4 --steps 24 > $O/f32_layers_c2.csv 2> $O/f32.err; grep "^#\|pool" $O/f32_layers_c2.csv | cut -c1-200
date
