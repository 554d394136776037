#!/bin/bash
# round 3, call c: per-launch parity on the C2 / C3 / C5 plans (fused C3 included) + golden tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py -m gpu -q -s --timeout 900 -p no:cacheprovider -k "every_conv_launch or golden" > $O/pytest.log 2>&1
echo "rc $?"; grep -v "^$" $O/pytest.log | tail -60 | cut -c1-300
