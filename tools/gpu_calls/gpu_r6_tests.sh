#!/bin/bash
# round 6: the factory coverage tests (fp32 mode end to end, per-launch parity of every factory's 16-bit plan)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06l}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests/test_factories_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider > $O/pytest_factories.log 2>&1
echo "factories rc $?"; grep -v "^$" $O/pytest_factories.log | tail -25 | cut -c1-250
timeout 1800 python -m pytest tests/test_parity_gpu.py -m gpu -q -s --timeout 900 -p no:cacheprovider -k "every_conv_launch" > $O/pytest_parity.log 2>&1
echo "parity rc $?"; grep "layer outputs\|passed\|failed\|Error\|assert" $O/pytest_parity.log | tail -30 | cut -c1-250
