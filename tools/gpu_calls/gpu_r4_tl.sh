#!/bin/bash
# round 4, call tl: kernel timeline of the pipelined C2 bench (four batches in flight): how much of the wall time has 0 / 1 / 2 / 3+ kernels resident, per-stream gaps
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --repeats 1 --no-cpu-baseline > /tmp/b_tl.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04tl
grep '^{"metric' /tmp/b_tl.log | tail -1 | cut -c1-300
DB=$(find /tmp/prof_tl -name "*.db" | head -1)
ls -la $DB
python tools/timeline_concurrency.py $DB 30 | tee gpurun_out/r04tl/concurrency.txt
python - <<PY
import sqlite3
cur = sqlite3.connect("$DB").cursor()
rows = list(cur.execute("select name, start, end, queue_id from kernels order by start"))
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - 6_000_000]
with open("gpurun_out/r04tl/last6ms.tsv", "w") as f:
    for n, s, e, q in rows:
        f.write(f"{q}\t{(s - rows[0][1]) / 1e3:.1f}\t{(e - s) / 1e3:.1f}\t{n.split('(')[0].replace('void ymi::', '')[:60]}\n")
PY
