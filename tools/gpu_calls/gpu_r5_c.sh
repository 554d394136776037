#!/bin/bash
# round 5, call C: software-pipelined fp32 tiles (211-216) -- op tests, all twelve tiles timed on every conv of the C2 plan, and the global-canvas GPU test
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05c
mkdir -p $O
date
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "f32" -p no:cacheprovider > $O/pytest_f32_ops.log 2>&1; echo "f32 ops rc $?"; tail -3 $O/pytest_f32_ops.log | cut -c1-300
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "global_canvas" -p no:cacheprovider > $O/pytest_canvas.log 2>&1; echo "canvas rc $?"; tail -15 $O/pytest_canvas.log | cut -c1-300
date
timeout 900 python tools/f32_layer_profile.py --config c2 --tune --depth 4 --steps 24 --json $O/f32_c2_tune.json > $O/f32_layers_c2_tune.csv 2> $O/f32_layers_c2_tune.err; grep "^#" $O/f32_layers_c2_tune.csv | cut -c1-700; tail -3 $O/f32_layers_c2_tune.err
date
