#!/bin/bash
# round 3, call z11: texture-path and L2 counters of the short-K 3x3 kernels with and without a shortcut (what the extra 5 600 cycles per tile are)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03z11
mkdir -p $O
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
i=$((i+1))
for res in 0 1; do
(cd /tmp && RES=$res TILES=93,132 timeout 100 rocprofv3 --pmc $set -d /tmp/pmc_${i}_$res -o r -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 8,64,64,320,320,3,1,1 > /tmp/pmc_${i}_$res.log 2>&1) || echo "pass $i res $res failed" >> $O/pmc.txt
db=$(find /tmp/pmc_${i}_$res -name "*.db" | head -1)
[ -n "$db" ] && python - "$db" "$set" $res >> $O/pmc.txt <<'PY'
import sqlite3, sys, collections
db, names, res = sys.argv[1], sys.argv[2], sys.argv[3]
cur = sqlite3.connect(db).cursor()
acc = collections.defaultdict(list)
try:
    for did, k, n, v, st in cur.execute("select dispatch_id, kernel_name, counter_name, value, start from counters_collection order by start, dispatch_id"):
        if "conv" in k: acc[(k[:70], n)].append(v)
    for (k, n), v in sorted(acc.items()):
        print(f"RES={res} {k:70s} {n:30s} mean {sum(v) / len(v):16.1f}  (x{len(v)})")
except Exception as e:
    print("query failed", e)
PY
done; done
cat $O/pmc.txt
