#!/bin/bash
# The scaling curve of BASELINE.json's metric in one command, on a node with 8 MI355X (none was available to rounds 1-3: the multi-GPU path is
# covered by the world-2 gloo tests, the one-rank RCCL test and the two-ranks-on-one-GPU control-flow test only).
#   bash tools/scale_8gpu.sh [steps] [warmup]    -> gpurun_out/scale/n{1,2,4,8}.json + a table (img/s, efficiency vs N x the 1-GPU figure)
# Each run is exactly what the driver launches: one rank per GPU over RCCL, weak scaling (32 images per GPU), slab all-gather per batch.
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${1:-60}; WARM=${2:-10}
O=gpurun_out/scale; mkdir -p $O
for N in 1 2 4 8; do
  if [ $N -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline > $O/n$N.log 2>&1
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps $STEPS --warmup $WARM > $O/n$N.log 2>&1
  fi
  grep '^{"metric' $O/n$N.log | tail -1 > $O/n$N.json
done
python - <<'P'
import json
base = None
print("N  img/s      ms/step  efficiency  second_rounds")
for n in (1, 2, 4, 8):
    try:
        d = json.loads(open(f"gpurun_out/scale/n{n}.json").read())
    except Exception as e:
        print(n, "failed:", e); continue
    base = base or d["value"]
    print(f"{n}  {d['value']:9.1f}  {d['ms_per_step']:7.4f}  {d['value'] / (n * base):10.3f}  {d['config'].get('gather_second_rounds_rank0')}")
P
