#!/bin/bash
# round 6: the strip kernel on / off on ONE box (YOLORT_AMD_C3_TILE=0 = the separate launches of round 5), alternating: bench c2 lines + the serial per-launch sums
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06onoff}
O=gpurun_out/$TAG
mkdir -p $O
for rep in 1 2; do
  for on in 0 1; do
    YOLORT_AMD_C3_TILE=$on timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2_tile${on}_$rep.log 2>&1
    grep '^{"metric' $O/bench_c2_tile${on}_$rep.log | tail -1 > $O/bench_c2_tile${on}_$rep.json
    python - <<PY >> $O/onoff.txt
import json
d=json.loads(open("$O/bench_c2_tile${on}_$rep.json").read())
print("YOLORT_AMD_C3_TILE=$on run $rep: value", d["value"], "img/s  ms_per_step", d["ms_per_step"], " serial conv ms", d["roofline"]["serial"]["conv_ms_per_step"], " roofline.frac", d["roofline"]["frac"], " launches", d["roofline"]["launches_per_step"])
PY
  done
done
for on in 0 1; do
  (cd /tmp && YOLORT_AMD_C3_TILE=$on timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t$on -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_t$on.json > /tmp/ps_t$on.log 2>&1)
  db=$(find /tmp/prof_t$on -name "*.db" | head -1)
  python tools/layer_table.py --ops $O/ops_t$on.json --stats $db > $O/layer_table_c2_tile$on.csv 2>> $O/err.log
  echo "YOLORT_AMD_C3_TILE=$on: $(grep '^# conv stack' $O/layer_table_c2_tile$on.csv)" >> $O/onoff.txt
done
cat $O/onoff.txt
