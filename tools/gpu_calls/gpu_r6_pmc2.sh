#!/bin/bash
# round 6: instruction-cache counters of the strip kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06j}
O=gpurun_out/$TAG
mkdir -p $O
CASES=${CASES:-32,40,40,256,128,1,0 32,80,80,256,64,1,0}
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && REPS=5 timeout 120 rocprofv3 --pmc $set -d /tmp/pmc_$i -o r -- python $GRAFT_REPO_ROOT/tools/c3t_run.py $CASES > /tmp/pmc_$i.log 2>&1) || echo "pmc pass $i failed"
  tail -1 /tmp/pmc_$i.log | cut -c1-200
done
python tools/pmc_kernel.py c3_tile $(for i in 1 2 3; do find /tmp/pmc_$i -name "*.db" | head -1; done) > $O/pmc_c3t_icache.txt 2>&1
cat $O/pmc_c3t_icache.txt
