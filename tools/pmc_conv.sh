#!/bin/bash
# PMC sweep of one conv case (tuning aid). usage: tools/pmc_conv.sh "<case>" <tile> <outdir>
CASE="$1"; TILE="$2"; OUT="$3"
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_BUSY_sum"; do
  i=$((i+1))
  TILES=$TILE rocprofv3 --pmc $set -d $OUT/p$i -o r -- python tools/conv_bench.py "$CASE" > $OUT/p$i.log 2>&1
done
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/p*/r_results.db")):
    cur = sqlite3.connect(db).cursor()
    cols=[d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    nm="kernel_name" if "kernel_name" in cols else "name"
    for name, cn, n, avg in cur.execute(f"select {nm}, counter_name, count(*), avg(value) from counters_collection where {nm} like '%conv_igemm%' group by {nm}, counter_name"):
        print(f"{cn:45s} n={n:3d} avg={avg:16.1f}  {name[:70]}")
PY
