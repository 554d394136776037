cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-cpu-baseline > /tmp/b_s.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find /tmp/prof_s -name "*.db" | head -1) > gpurun_out/r01d_rocprof_summary.csv
grep '^{"metric' /tmp/b_s.log | tail -1 > gpurun_out/r01d_bench_under_rocprof.json
cut -c1-200 gpurun_out/r01d_bench_under_rocprof.json
head -5 gpurun_out/r01d_rocprof_summary.csv | cut -c1-50,180-260
