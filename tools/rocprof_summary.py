"""Turns rocprofv3's rocpd SQLite output into small text/CSV summaries for profiles/.

usage: python tools/rocprof_summary.py <stats.db> [--pmc <fetch.db> <write.db>] > profiles/rNN_summary.txt
"""
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)")
    print("name,calls,total_us,avg_us,percent")
    for name, calls, tot, avg, pct in rows:
        print(f'"{name}",{calls},{tot:.1f},{avg:.3f},{pct:.2f}')


def pmc(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    q = f"select {name_col}, counter_name, count(*), avg(value), sum(value) from counters_collection group by {name_col}, counter_name order by sum(value) desc"
    print(f"# rocprofv3 --pmc {counter}: per-kernel averages (raw counter units; FETCH_SIZE/WRITE_SIZE are KB)")
    print("name,counter,dispatches,avg_value,sum_value")
    for name, cn, n, avg, tot in cur.execute(q):
        print(f'"{name}",{cn},{n},{avg:.2f},{tot:.1f}')


if __name__ == "__main__":
    kernel_stats(sys.argv[1])
    if "--pmc" in sys.argv:
        i = sys.argv.index("--pmc")
        for db in sys.argv[i + 1:]:
            print()
            pmc(db, db)
