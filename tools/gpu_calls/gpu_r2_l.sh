#!/bin/bash
# round 2, GPU call L: where the head's 95 us go (conv only / first K step only / neither), gather test
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02l
mkdir -p $O
timeout 600 python -m pytest tests/test_boundary_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gather" > $O/tests.log 2>&1; tail -3 $O/tests.log; grep -n "mismatch image\|total detections" $O/tests.log | head
for hs in 1 0; do for dbg in 0 4 16 20; do
(cd /tmp && rm -rf /tmp/prof_t && YOLORT_AMD_HEAD_SPLIT=$hs YOLORT_AMD_HEAD_DEBUG=$dbg timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 6 --score-thresh 0.999 --ops $GRAFT_REPO_ROOT/$O/ops_c2.json > /tmp/ps_t.log 2>&1)
db=$(find /tmp/prof_t -name "*.db" | head -1)
python tools/layer_table.py --ops $O/ops_c2.json --stats $db > $O/lt.csv 2>> $O/err.log
echo "split $hs debug $dbg: $(grep -E '^(0|47),' $O/lt.csv | cut -d, -f1,6-8 | tr '\n' ' ' | cut -c1-200)"
done; done
tail -3 $O/err.log
