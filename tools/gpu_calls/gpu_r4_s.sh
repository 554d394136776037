#!/bin/bash
# round 4, call s: the C3 and C5 bench lines again (the first measurement set's C3 line died of memory: the fp32-mode throughput leg built four fp32 instances of yolov5m bs 64)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04z
mkdir -p $O
for c in c3 c5; do
  timeout 900 python bench.py --config $c > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; cut -c1-260 $O/bench_$c.json; tail -2 $O/bench_$c.log | cut -c1-300 | grep -v metric
done
