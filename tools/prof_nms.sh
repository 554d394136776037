cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_n -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline > /tmp/b_n.log 2>&1
python - <<'PY'
import sqlite3, glob
db=glob.glob('/tmp/prof_n/**/*.db', recursive=True)[0]
con=sqlite3.connect(db); cur=con.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt=[t for t in tabs if 'kernel' in t.lower()]
print(kt[:20])
v='kernels' if 'kernels' in tabs else kt[0]
cols=[d[1] for d in cur.execute(f"pragma table_info({v})")]
print(v, cols)
nm='name' if 'name' in cols else 'kernel_name'
rows=list(cur.execute(f"select {nm}, start, end from {v} order by start"))
t0=rows[0][1]
for n,s,e in rows:
    if 'nms_segments' in n or 'sort_image' in n or 'head_decode' in n or 'select_prefix' in n:
        if (e-s) > 5e5 or "XX" in n: print(f"{(s-t0)/1e6:10.3f} ms  dur {(e-s)/1e3:10.1f} us  {n[:40]}")
PY
