#!/bin/bash
# round 3, call g: counted vmcnt wait in the fused stem kernel: tests, then A/B (YOLORT_AMD_SB_DEBUG=1 = wait for everything)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
timeout 900 python -m pytest tests/test_c3_fused_gpu.py tests/test_parity_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "fused_stem or (every_conv_launch and s_r60)" > $O/pytest.log 2>&1
rc=$?; echo "tests rc $rc"; grep -v "^$" $O/pytest.log | tail -6 | cut -c1-300
[ $rc -ne 0 ] && exit 0
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl: c2', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], r['launches_per_step'])"
}
for rep in 1 2 3; do
run "vmcnt(0)" YOLORT_AMD_SB_DEBUG=1
run "counted wait" YOLORT_AMD_SB_DEBUG=0
done
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 50 --per-op $O/perop.json > $O/bench_perop.log 2>&1
python - <<'P'
import json
for r in json.load(open('gpurun_out/r03g/perop.json'))[:4]:
    print(r['name'], round(r['ms']*1e3,1), 'us')
P
