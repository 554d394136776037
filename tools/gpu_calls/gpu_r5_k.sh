#!/bin/bash
# round 5, call K: is call J's 22.2 k img/s (conv stack 1.66 ms) the box or the build?  the bench again + the serial per-kernel table, and the legacy per-launch tests after their count fix
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05k}
mkdir -p $O
rocm-smi --showclocks --showpower 2>/dev/null | head -20 > $O/smi_before.txt
timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-200 $O/bench_c2.json
python - <<PY
import json
d=json.load(open("$O/bench_c2.json"))
print("serial", d["roofline"]["serial"]["conv_ms_per_step"], "host", d["roofline"].get("timed_region",{}).get("host_enqueue_ms_per_step_rank0"))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_c2.json > /tmp/ps_c2.log 2>&1)
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python tools/layer_table.py --ops $O/ops_c2.json --stats $db > $O/layer_table_c2.csv 2>> $O/err.log; grep "^# conv stack" $O/layer_table_c2.csv
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "r40 or r31" 2>&1 | tail -3
