#!/bin/bash
# round-2 final measurement set at HEAD: targeted tests, bench c2/c3/c5, rocprofv3 stats + layer tables, PMC passes (c2), letterbox modes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r02z}
O=gpurun_out/$TAG
mkdir -p $O
date
# (targeted tests ran in the first pass of this script, r02z)
date
for c in c2 c3 c5; do
  timeout 500 python bench.py --config $c > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; cut -c1-200 $O/bench_$c.json
done
date
for d in 0 1 2 3; do YOLORT_AMD_LB_DEBUG=$d timeout 120 python tools/letterbox_bench.py c3 30 2>&1 | grep "^letterbox"; done | tee $O/lb_modes.txt
for cfg in c2 c3 c5; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/rocprof_summary.py $db > $O/rocprof_summary_$cfg.csv 2>> $O/err.log
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
  tail -14 $O/layer_table_$cfg.csv | head -3
done
date
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set -d /tmp/pmc_$i -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 > /tmp/pmc_$i.log 2>&1)
done
dbs=$(for i in 1 2 3 4; do find /tmp/pmc_$i -name "*.db" | head -1; done)
python tools/layer_table.py --ops $O/ops_c2.json --stats $(find /tmp/prof_c2 -name "*.db" | head -1) --pmc $dbs > $O/layer_table_c2_pmc.csv 2>> $O/err.log
tail -3 $O/layer_table_c2_pmc.csv | cut -c1-300; tail -5 $O/err.log
date
