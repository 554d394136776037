#!/bin/bash
# round 4, call a: the whole GPU suite at the round's starting point (+ the hook regression test), the default bench line, and the GEMM yardstick (hipBLASLt via torch.mm)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -6 $O/pytest_all.log | cut -c1-300
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc $?"; cut -c1-400 $O/bench_c2.json
for cfg in c2 c5; do
  timeout 300 python tools/gemm_yardstick.py $cfg 2>/dev/null > $O/gemm_yardstick_$cfg.txt; echo "yardstick $cfg rc $?"
done
cat $O/gemm_yardstick_c2.txt | cut -c1-200
