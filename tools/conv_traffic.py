"""profiles/conv_traffic.json (the `roofline.traffic` source of bench.py) from a per-layer PMC table (tools/layer_table.py --pmc ...):
HBM bytes of the conv + fused-head launches of one C2 step = sum over launches of FETCH_SIZE x 2 (gfx950 correction,
MI355X_MICROARCH.md HBM section; the counter is in KiB-like units of 1000 B as rocprofv3 reports it) + WRITE_SIZE.

    python tools/conv_traffic.py profiles/r02z_layer_table_c2_pmc.csv > profiles/conv_traffic.json
"""
import csv
import json
import sys

path = sys.argv[1]
rows = [r for r in csv.reader(l for l in open(path) if not l.startswith("#"))]
ix = {n: i for i, n in enumerate(rows[0])}
fetch = sum(float(r[ix["FETCH_SIZE"]]) for r in rows[1:] if r[ix["FETCH_SIZE"]])
write = sum(float(r[ix["WRITE_SIZE"]]) for r in rows[1:] if r[ix["WRITE_SIZE"]])
print(json.dumps({
    "fetch_mb_per_step_raw": round(fetch / 1e3, 1),
    "fetch_mb_per_step_corrected": round(2 * fetch / 1e3, 1),
    "write_mb_per_step": round(write / 1e3, 1),
    "launches": len(rows) - 1,
    "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "source": f"{path}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/profile_serial.py --config c2 (one batch in flight, pinned tile table), "
              "conv + pool + fused-head launches, mean of 8 steps",
    "workload": "yolov5_darknet_pan_s_r60 fp16 bs=32 640x640",
}, indent=1))
