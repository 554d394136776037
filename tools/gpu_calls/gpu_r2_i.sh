#!/bin/bash
# round 2, GPU call I: per-layer table with PMC (fixed matching), LDS-floor / pipeline-depth experiment, new tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02i
mkdir -p $O
date
timeout 600 python -m pytest tests/test_boundary_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gather or chained_1x1 or stream" > $O/tests.log 2>&1; tail -3 $O/tests.log
for fl in 0 54 81; do for pd in 4 6; do
YOLORT_AMD_LDS_FLOOR_KB=$fl YOLORT_AMD_PIPELINE=$pd timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 100 > $O/fl_${fl}_$pd.log 2>&1; python -c "
import json
d=json.loads([l for l in open('$O/fl_${fl}_$pd.log') if l.startswith('{\"metric')][-1]); r=d['roofline']
print('lds floor $fl KB, pipeline $pd:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; conv excl', r['conv_ms_per_step'])
"
done; done
date
for cfg in c2 c3 c5; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/rocprof_summary.py $db > $O/rocprof_summary_$cfg.csv 2>> $O/err.log
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
done
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --pmc $set -d /tmp/pmc_$i -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 > /tmp/pmc_$i.log 2>&1)
done
dbs=$(for i in 1 2 3 4; do find /tmp/pmc_$i -name "*.db" | head -1; done)
python tools/layer_table.py --ops $O/ops_c2.json --stats $(find /tmp/prof_c2 -name "*.db" | head -1) --pmc $dbs > $O/layer_table_c2_pmc.csv 2>> $O/err.log
tail -12 $O/layer_table_c2_pmc.csv | cut -c1-200; tail -5 $O/err.log
date
