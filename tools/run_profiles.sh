#!/bin/bash
# end-of-round measurement set: bench (with cpu baseline), kernel-trace stats, FETCH/WRITE PMC passes (separate runs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python bench.py --steps 100 --warmup 10 > gpurun_out/r01d_bench.log 2>&1; grep '^{"metric' gpurun_out/r01d_bench.log | tail -1 > gpurun_out/r01d_bench.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-cpu-baseline > /tmp/b_s.log 2>&1)
python tools/rocprof_summary.py $(find /tmp/prof_s -name "*.db" | head -1) > gpurun_out/r01d_rocprof_summary.csv
grep '^{"metric' /tmp/b_s.log | tail -1 > gpurun_out/r01d_bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && YOLORT_AMD_AUTOTUNE=0 timeout 400 rocprofv3 --pmc $c -d /tmp/prof_$c -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline > /tmp/b_$c.log 2>&1)
done
python - <<'PY' > gpurun_out/r01d_conv_traffic.txt
import sqlite3, glob
tot={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    dbs=glob.glob(f"/tmp/prof_{c}/**/*.db", recursive=True)
    if not dbs: print(c,"no db"); continue
    cur=sqlite3.connect(dbs[0]).cursor()
    cols=[d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    nm="kernel_name" if "kernel_name" in cols else "name"
    rows=list(cur.execute(f"select {nm}, count(*), sum(value) from counters_collection where counter_name='{c}' group by {nm}"))
    conv=sum(v for n,k,v in rows if 'conv' in n)
    grp=sum(k for n,k,v in rows if 'conv_head_decode_group' in n)   # one grouped head launch per forward pass (else three per-level launches)
    nsteps=max(1, grp if grp else sum(k for n,k,v in rows if 'conv_head_decode' in n)//3)
    allk=sum(v for n,k,v in rows)
    tot[c]=(conv,allk,nsteps)
    print(f"# {c}: conv kernels {conv/1024:.1f} MB total over the run, all kernels {allk/1024:.1f} MB (counter unit KB)")
    for n,k,v in sorted(rows,key=lambda r:-r[2])[:12]: print(f"{c},{k},{v/1024:.1f} MB,{n[:90]}")
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    steps=tot["FETCH_SIZE"][2]
    f=tot["FETCH_SIZE"][0]/1024/steps; w=tot["WRITE_SIZE"][0]/1024/tot["WRITE_SIZE"][2]
    print(f"conv kernels (incl. fused head), autotune off, {steps} forward passes in the run (timed steps + the exclusive-conv and parity passes): FETCH_SIZE {f:.1f} MB/step raw ({2*f:.1f} MB with the gfx950 x2 correction), WRITE_SIZE {w:.1f} MB/step")
PY
cat gpurun_out/r01d_conv_traffic.txt | tail -3
cut -c1-400 gpurun_out/r01d_bench.json
