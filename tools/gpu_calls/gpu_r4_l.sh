#!/bin/bash
# round 4, call l: the upsampled copy (nn.Upsample folded into its producer) written as whole rows through the row-transposed-store tiles: op tests, then same-box A/Bs of C2 with the
# two up2 layers (pan.inner_blocks.1 at 20^2, pan.inner_blocks.4 at 40^2) on tiles 144 / 145 / 142 and 143 / 142 instead of the table's 65 / 66
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04l
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "upsampled_second_output" -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300 | tee $O/tests_up2.txt
run() { lbl=$1; shift
  env "$@" timeout 300 python bench.py --config c2 --no-cpu-baseline --per-op $O/perop_$lbl.json 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lbl', d['value'], d['ms_per_step'], r['serial']['conv_ms_per_step'], r['frac'], d['repeats']['spread_pct'])"
  python -c "
import json
ops=json.load(open('$O/perop_$lbl.json'))
print('   ', [(o['name'].split('.')[-1], o['tile'], round(o['ms']*1e3,1)) for o in ops if 'inner_blocks.1' in o['name'] or 'inner_blocks.4' in o['name']])"
}
for rep in 1 2; do
run table A=1 | tee -a $O/ab_up2.txt
for v in tp_a tp_b tp_c tp_d; do
run $v YOLORT_AMD_TILE_TABLE_PATH=$PWD/tools/_ab/tiles_$v.json | tee -a $O/ab_up2.txt
done
done
