#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  (cd /tmp && TILES=93,91,92 timeout 300 rocprofv3 --pmc $set -d /tmp/pmc_$i -o r -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 32,128,128,40,40,3,1,1 > /tmp/pmc_$i.log 2>&1)
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  python - "$db" <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = list(cur.execute("select kernel_name, counter_name, value from counters_collection"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v in rows:
    if "halo8" in k: acc[k.split("(")[0][-40:]][c].append(v)
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
