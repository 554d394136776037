#!/bin/bash
# round 3, first call: the kernels written blind at the end of round 2 (no GPU minutes were left).
#   1. the gated tests of the fused C3 launch (bit-identity with the separate launches, torch fp32, end to end)
#   2. only if they pass: same-box A/B of the C2 bench, three interleaved repeats, plus the per-op table with the knob on
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
YOLORT_AMD_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_c3_fused_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "not row_transposed" > $O/pytest_c3fused.log 2>&1
rc=$?
tail -15 $O/pytest_c3fused.log
YOLORT_AMD_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_c3_fused_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "row_transposed" > $O/pytest_tp.log 2>&1
rctp=$?
tail -15 $O/pytest_tp.log
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 200 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lbl: c2', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d.get('parity'))"
}
if [ $rc -eq 0 ]; then
for rep in 1 2; do
run "separate launches" YOLORT_AMD_FUSE_C3=0
run "fused C3" YOLORT_AMD_FUSE_C3=1
done
YOLORT_AMD_FUSE_C3=1 timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 50 --per-op $O/perop_c3fused.json > $O/bench_c3fused.log 2>&1
python - <<'P'
import json
for r in json.load(open('gpurun_out/r03a/perop_c3fused.json'))[:8]:
    print(r)
P
else echo "fused C3: tests FAILED (rc $rc) -- no A/B"; fi
[ $rctp -ne 0 ] && { echo "TP tiles: tests FAILED (rc $rctp) -- no re-tune"; exit 0; }
# 3. row-transposed-store tiles (141-145, 151-155): offered to the tuner only under YOLORT_AMD_TUNE_TP=1; re-tune C2 with them (and with the
#    streaming kernel's row stores, which the committed table predates), then A/B the new table against the committed one
YOLORT_AMD_TUNE_TP=1 timeout 420 python tools/tune_tiles.py --out $O/tiles_tp.json yolov5_darknet_pan_s_r60:fp16:32:640 > $O/tune_tp.log 2>&1; tail -2 $O/tune_tp.log | cut -c1-200
for rep in 1 2; do
run "committed table" A=1
run "re-tuned with TP tiles" YOLORT_AMD_TILE_TABLE_PATH=$PWD/$O/tiles_tp.json
done
python - <<'P'
import json
old=json.load(open('yolort_amd/data/tiles_gfx950.json'))['tiles']; new=json.load(open('gpurun_out/r03a/tiles_tp.json'))['tiles']
for k,v in new.items():
    if old.get(k)!=v: print(k, old.get(k), '->', v)
P
