#!/bin/bash
# round 4, call t: two cheap same-box A/Bs on C2 (and the first on C5)
#   (1) s_setprio around the MFMA groups of the 8-wave kernels (conv_halo8 / conv_igemm8): tools/_ab/libyolort_amd_prio{1,3}.so = the shipped sources + -DYMI_SETPRIO=1 / 3
#   (2) C3.cv1|cv2 with the first Bottleneck's 1x1 chained at hidden width 128 (the three 40x40 C3s of yolov5s), carried by tile 116 / 78 (YOLORT_AMD_CHAIN128)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04t
mkdir -p $O
run() { cfg=$1; lbl=$2; shift; shift
  env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline --per-op $O/perop_${cfg}_$lbl.json 2>$O/err_${cfg}_$lbl.txt | grep '^{"metric' > $O/line_${cfg}_$lbl.json
  python - <<PY
import json
try:
    d = json.loads(open('$O/line_${cfg}_$lbl.json').readline()); r = d['roofline']
    par = d.get('parity', {})
    print('$cfg', '$lbl', 'img/s', d['value'], 'ms/step', d['ms_per_step'], 'serial conv ms', r['serial']['conv_ms_per_step'], 'frac', r['frac'], 'spread %', d['repeats']['spread_pct'],
          'parity', {k: (v['production_fp16']['paired'], v['production_fp16']['ref_dets'], v['production_fp16']['min_iou']) for k, v in par.items() if isinstance(v, dict) and 'production_fp16' in v})
except Exception as e:
    print('$cfg', '$lbl', 'FAILED', e, open('$O/err_${cfg}_$lbl.txt').read()[-600:])
PY
}
# the chained launch at width 128 against the oracle's layers (every conv launch of the plan, incl. the chained 1x1's output)
YOLORT_AMD_CHAIN128=116 timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "every_conv_launch" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests_chain116.txt
for rep in 1 2; do
  run c2 base A=1 | tee -a $O/ab_c2.txt
  run c2 prio1 YOLORT_AMD_LIB=$PWD/tools/_ab/libyolort_amd_prio1.so | tee -a $O/ab_c2.txt
  run c2 prio3 YOLORT_AMD_LIB=$PWD/tools/_ab/libyolort_amd_prio3.so | tee -a $O/ab_c2.txt
  run c2 chain116 YOLORT_AMD_CHAIN128=116 | tee -a $O/ab_c2.txt
done
run c2 chain78 YOLORT_AMD_CHAIN128=78 | tee -a $O/ab_c2.txt
run c5 base A=1 | tee -a $O/ab_c5.txt
run c5 prio1 YOLORT_AMD_LIB=$PWD/tools/_ab/libyolort_amd_prio1.so | tee -a $O/ab_c5.txt
run c5 base2 A=1 | tee -a $O/ab_c5.txt
# per-op view of what changed (us, serial): ops whose time differs by > 3 % between base and a variant
python - <<'PY' | tee gpurun_out/r04t/perop_diff.txt
import json, glob, os
O = 'gpurun_out/r04t'
for cfg in ('c2', 'c5'):
    try:
        base = json.load(open(f'{O}/perop_{cfg}_base.json'))
    except Exception as e:
        print(cfg, 'no base', e); continue
    for f in sorted(glob.glob(f'{O}/perop_{cfg}_*.json')):
        lbl = os.path.basename(f)[len(f'perop_{cfg}_'):-5]
        if lbl == 'base':
            continue
        ops = json.load(open(f))
        tb, tv = sum(o['ms'] for o in base) * 1e3, sum(o['ms'] for o in ops) * 1e3
        print(f'{cfg} {lbl}: per-op sum {tv:.1f} us vs base {tb:.1f} us ({len(ops)} vs {len(base)} launches)')
        bn = {o['name']: o for o in base}
        for o in ops:
            b = bn.get(o['name'])
            if b is None:
                print(f"    {o['name']:55s} tile {o['tile']:4d} {o['ms']*1e3:7.1f} us  (no such launch in base)")
            elif abs(o['ms'] - b['ms']) > 0.03 * b['ms']:
                print(f"    {o['name']:55s} tile {o['tile']:4d} {o['ms']*1e3:7.1f} us  base tile {b['tile']:4d} {b['ms']*1e3:7.1f} us")
        for n in bn:
            if n not in {o['name'] for o in ops}:
                print(f"    {n:55s} only in base: tile {bn[n]['tile']:4d} {bn[n]['ms']*1e3:7.1f} us")
PY
