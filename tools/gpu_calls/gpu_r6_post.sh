#!/bin/bash
# round 6: the post-process range as a second captured graph -- boundary tests, host time of one submitted batch, bench A/B (YOLORT_AMD_POST_GRAPH=0 / 1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06v}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_e2e_gpu.py -m gpu -q -x --timeout 800 -p no:cacheprovider > $O/pytest_boundary.log 2>&1
echo "boundary rc $?"; tail -4 $O/pytest_boundary.log | cut -c1-300
timeout 300 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; grep -v amdgpu.ids $O/host_overhead.txt | tail -6
for pg in 0 1 0 1; do
  YOLORT_AMD_POST_GRAPH=$pg timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2_pg$pg.log 2>&1
  grep '^{"metric' $O/bench_c2_pg$pg.log | tail -1 > $O/bench_c2_pg$pg.json
  python - <<PY
import json
d=json.loads(open("$O/bench_c2_pg$pg.json").read())
print("post_graph=$pg", d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step_rank0"], d["config"]["serving_mode_rank0"])
PY
done
