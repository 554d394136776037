#!/bin/bash
# round 2, GPU call A: tile table, full GPU test suite, bench lines for c2/c3/c5, rocprofv3 kernel trace + PMC passes (serial driver)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
echo "== tune tiles" ; date
timeout 900 python tools/tune_tiles.py --out $O/tiles_gfx950.json > $O/tune.log 2>&1; tail -3 $O/tune.log
[ -s $O/tiles_gfx950.json ] && cp $O/tiles_gfx950.json yolort_amd/data/tiles_gfx950.json
echo "== tests" ; date
timeout 1800 python -m pytest tests -m gpu -q -s --timeout 900 -p no:cacheprovider > $O/tests.log 2>&1; grep -v "^$" $O/tests.log | tail -60
echo "== bench" ; date
timeout 600 python bench.py > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-600 $O/bench_c2.json
timeout 600 python bench.py --config c3 > $O/bench_c3.log 2>&1; grep '^{"metric' $O/bench_c3.log | tail -1 > $O/bench_c3.json; cut -c1-300 $O/bench_c3.json
timeout 600 python bench.py --config c5 > $O/bench_c5.log 2>&1; grep '^{"metric' $O/bench_c5.log | tail -1 > $O/bench_c5.json; cut -c1-300 $O/bench_c5.json
echo "== rocprof" ; date
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
tail -3 $O/layer_table_c2_pmc.csv; tail -5 $O/err.log
date
