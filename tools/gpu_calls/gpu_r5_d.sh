#!/bin/bash
# round 5, call D: what bounds the fp32 tiles at 0.53 matrix-pipe busy?  Single-wave blocks (no barrier), four-deep rings, one block per CU (LDS floor), every form on every conv of C2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05d
mkdir -p $O
date
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "f32 and bit_for_bit" -p no:cacheprovider > $O/pytest_f32_ops.log 2>&1; echo "f32 ops rc $?"; tail -3 $O/pytest_f32_ops.log | cut -c1-300
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "global_canvas" -p no:cacheprovider > $O/pytest_canvas.log 2>&1; echo "canvas rc $?"; tail -5 $O/pytest_canvas.log | cut -c1-300
date
YOLORT_AMD_F32_TUNE_TILES=201,202,211,212,221,222,207,217,218,219,206 timeout 900 python tools/f32_layer_profile.py --config c2 --tune --depth 4 --steps 16 > $O/f32_tune_experiments.csv 2> $O/f32_tune_experiments.err; grep "^#" $O/f32_tune_experiments.csv | cut -c1-700; tail -3 $O/f32_tune_experiments.err
date
YOLORT_AMD_LDS_FLOOR_KB=81 YOLORT_AMD_F32_TUNE_TILES=201,202,211,212 timeout 900 python tools/f32_layer_profile.py --config c2 --tune --depth 1 --steps 8 > $O/f32_tune_one_block_per_cu.csv 2> $O/f32_tune_one_block_per_cu.err; grep "^#" $O/f32_tune_one_block_per_cu.csv | cut -c1-400
date
