#!/bin/bash
# round 6: SQ / LDS counters of the strip kernel on single C3 blocks
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06e}
O=gpurun_out/$TAG
mkdir -p $O
CASES=${CASES:-32,40,40,256,128,1,0 32,80,80,256,64,1,0}
timeout 200 python tools/c3t_run.py $CASES 2>&1 | tail -3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && REPS=5 timeout 300 rocprofv3 --pmc $set -d /tmp/pmc_$i -o r -- python $GRAFT_REPO_ROOT/tools/c3t_run.py $CASES > /tmp/pmc_$i.log 2>&1) || echo "pmc pass $i failed"
  tail -2 /tmp/pmc_$i.log | cut -c1-200
done
python tools/pmc_kernel.py c3_tile $(for i in 1 2 3; do find /tmp/pmc_$i -name "*.db" | head -1; done) > $O/pmc_c3t.txt 2>&1
cat $O/pmc_c3t.txt
