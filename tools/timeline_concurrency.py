"""Concurrency analysis of a rocprofv3 --kernel-trace database of a pipelined bench run (tuning aid).

    python tools/timeline_concurrency.py <trace.db> [window_ms]

Over the LAST `window_ms` milliseconds of the trace (the timed region of bench.py: several batches in flight on their own streams): the fraction of time with 0 / 1 / 2 / >= 3
kernels resident, the busy time per kernel family, and per stream (queue) the share of its own time spent between kernels (launch gaps / event waits)."""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
t_end = max(r[2] for r in rows)
t0 = t_end - int(window_ms * 1e6)
rows = [r for r in rows if r[1] >= t0]
print(f"{len(rows)} dispatches in the last {window_ms} ms; columns: {cols}")
ev = []
for r in rows:
    ev.append((r[1], 1))
    ev.append((r[2], -1))
ev.sort()
lvl, last, hist = 0, ev[0][0], defaultdict(int)
for t, d in ev:
    hist[min(lvl, 4)] += t - last
    last = t
    lvl += d
tot = sum(hist.values())
print("time share by number of kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
fam = defaultdict(int)
for r in rows:
    fam[r[0].split("(")[0].replace("void ymi::", "")[:48]] += r[2] - r[1]
print("kernel-time / wall (sum over kernels of duration, divided by the window):")
for k, v in sorted(fam.items(), key=lambda x: -x[1])[:14]:
    print(f"   {k:50s} {v / tot:6.3f}")
print("   total", round(sum(fam.values()) / tot, 3))
if qcol:
    per = defaultdict(list)
    for r in rows:
        per[r[3]].append((r[1], r[2]))
    for q, iv in sorted(per.items()):
        busy = sum(e - s for s, e in iv)
        span = iv[-1][1] - iv[0][0]
        gaps = sorted((iv[i + 1][0] - iv[i][1]) for i in range(len(iv) - 1))
        print(f"queue {q}: {len(iv)} kernels, busy {busy / span:.3f} of its span, median gap {gaps[len(gaps) // 2] / 1e3:.1f} us, p90 gap {gaps[int(0.9 * len(gaps))] / 1e3:.1f} us")
