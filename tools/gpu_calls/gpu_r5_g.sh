#!/bin/bash
# round 5, call G: the two-call C submit (ymi_plan_begin / ymi_plan_submit) + the C form of the weights walk: e2e / boundary / dist-free suites, host cost per batch with the A/B partners
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05g}
mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_boundary_gpu.py tests/test_configs_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest_e2e.log 2>&1; tail -6 $O/pytest_e2e.log | cut -c1-300
timeout 300 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; tail -5 $O/host_overhead.txt
YOLORT_AMD_C_SUBMIT=0 timeout 300 python tools/host_overhead.py > $O/host_overhead_torch_submit.txt 2>&1; tail -3 $O/host_overhead_torch_submit.txt
YOLORT_AMD_SIG_EXT=0 timeout 300 python tools/host_overhead.py > $O/host_overhead_python_walk.txt 2>&1; tail -3 $O/host_overhead_python_walk.txt
timeout 600 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-300 $O/bench_c2.json
python - <<PY
import json
d=json.load(open("$O/bench_c2.json"))
print("host_enqueue", d["roofline"].get("host_enqueue_ms_per_step_rank0"), "serving", d["roofline"].get("serving_mode_rank0"))
PY
