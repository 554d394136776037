#!/bin/bash
# round-6 measurement set at HEAD (TAG selects the output directory): the whole GPU suite + smoke, bench lines c2 (headline) / c2 fp32 / c3 / c5, the golden tests verbosely, rocprofv3 kernel
# statistics + per-layer tables of the serial driver (c2, c3, c5, c2 fp32), PMC passes for c2 (SQ counters; FETCH_SIZE / WRITE_SIZE in their own passes) and for the fp32 mode
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06z}
O=gpurun_out/$TAG
mkdir -p $O
date
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; tail -8 $O/pytest_all.log | cut -c1-300
timeout 400 python -m pytest tests/test_golden_gpu.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | cut -c1-800 > $O/golden_verbose.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
date
timeout 600 python bench.py --config c2 > $O/bench_c2.log 2>&1; grep '^{"metric' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-260 $O/bench_c2.json
timeout 600 python bench.py --config c2 --dtype fp32 > $O/bench_c2_fp32.log 2>&1; grep '^{"metric' $O/bench_c2_fp32.log | tail -1 > $O/bench_c2_fp32.json; cut -c1-260 $O/bench_c2_fp32.json
for c in c3 c5; do
  timeout 600 python bench.py --config $c > $O/bench_$c.log 2>&1; grep '^{"metric' $O/bench_$c.log | tail -1 > $O/bench_$c.json; cut -c1-260 $O/bench_$c.json
done
date
for cfg in c2 c3 c5; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config $cfg --steps 8 --ops $GRAFT_REPO_ROOT/$O/ops_$cfg.json > /tmp/ps_$cfg.log 2>&1)
  db=$(find /tmp/prof_$cfg -name "*.db" | head -1)
  python tools/rocprof_summary.py $db > $O/rocprof_summary_$cfg.csv 2>> $O/err.log
  python tools/layer_table.py --ops $O/ops_$cfg.json --stats $db > $O/layer_table_$cfg.csv 2>> $O/err.log
  grep "^# conv stack" $O/layer_table_$cfg.csv
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2f32 -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --fp32 --steps 8 > /tmp/ps_c2f32.log 2>&1)
python tools/rocprof_summary.py $(find /tmp/prof_c2f32 -name "*.db" | head -1) > $O/rocprof_summary_c2_fp32.csv 2>> $O/err.log; head -5 $O/rocprof_summary_c2_fp32.csv | cut -c1-160
timeout 300 python tools/f32_layer_profile.py --config c2 --depth 4 --steps 24 > $O/f32_layers_c2.csv 2>> $O/err.log; grep "^#" $O/f32_layers_c2.csv
date
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set -d /tmp/pmc_$i -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 > /tmp/pmc_$i.log 2>&1) || echo "pmc pass $i failed"
done
dbs=$(for i in 1 2 3 4 5; do find /tmp/pmc_$i -name "*.db" 2>/dev/null | head -1; done)
python tools/layer_table.py --ops $O/ops_c2.json --stats $(find /tmp/prof_c2 -name "*.db" | head -1) --pmc $dbs > $O/layer_table_c2_pmc.csv 2>> $O/err.log
python tools/conv_traffic.py $O/layer_table_c2_pmc.csv > $O/conv_traffic.json 2>> $O/err.log; head -6 $O/conv_traffic.json
tail -3 $O/layer_table_c2_pmc.csv | cut -c1-300
for cn in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $cn | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $cn -d /tmp/pmcf_$tag -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --fp32 --steps 8 > /tmp/pmcf_$tag.log 2>&1) || echo "fp32 pmc pass $tag failed"
done
python tools/pmc_sum.py --steps 11 --match conv_f32_pipe FETCH=$(find /tmp/pmcf_FETCH_SIZE -name "*.db" | head -1) WRITE=$(find /tmp/pmcf_WRITE_SIZE -name "*.db" | head -1) MFMA=$(find /tmp/pmcf_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" | head -1) > $O/conv_traffic_fp32.json 2>> $O/err.log; cat $O/conv_traffic_fp32.json | head -30
tail -5 $O/err.log
date
