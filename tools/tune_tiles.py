"""Regenerates yolort_amd/data/tiles_gfx950.json, the pinned per-(shape, dtype) convolution tile table (run on an MI355X).

    python tools/tune_tiles.py [--out PATH] [--merge] [config ...]     config = arch:dtype:batch:size[:dynamic]

Every conv launch of each configuration's plan is timed once per candidate tile on its real buffers (Plan._autotune_tile,
interleaved best-of-3 rounds); the winner per shape key is written to the table.  The product never times anything at
plan build (YOLORT_AMD_AUTOTUNE defaults to 0): the committed table makes tile choice -- and with it the K accumulation
order and the detections -- identical across processes, ranks and runs.
"""
import argparse
import json
import os
import sys

os.environ["YOLORT_AMD_AUTOTUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yolort_amd import engine  # noqa: E402
from yolort_amd.models import YOLOv5  # noqa: E402
from workloads.synth import synth_images, synth_weights  # noqa: E402

DEFAULT = ["yolov5_darknet_pan_s_r60:fp16:32:640", "yolov5_darknet_pan_n_r60:fp16:2:640", "yolov5_darknet_pan_m_r60:bf16:64:1280:dynamic",
           "yolov5_darknet_pan_l6_r60:fp16:8:1280"]
C3_SHAPES = [(1080, 1920), (720, 1280), (1920, 1080), (1080, 810), (960, 1280), (1281, 1279), (641, 480), (375, 500)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=DEFAULT)
    ap.add_argument("--out", default=engine.TILE_TABLE_PATH)
    ap.add_argument("--merge", action="store_true", help="keep the entries of an existing table that this run does not re-measure")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for cfg in args.configs:
        f = cfg.split(":")
        arch, dt, batch, size = f[0], {"fp16": torch.float16, "bf16": torch.bfloat16}[f[1]], int(f[2]), int(f[3])
        dynamic = len(f) > 4 and f[4] == "dynamic"
        kw = dict(size_divisible=64) if arch.endswith("6_r60") else {}
        m = YOLOv5(arch=arch, size=(size, size), score_thresh=0.25, **kw)
        m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0))
        m = m.to(dev).to(dt).eval()
        if dynamic:
            imgs = [synth_images(1, *C3_SHAPES[i % len(C3_SHAPES)], seed=1 + i)[0].to(dev).to(dt) for i in range(batch)]
        else:
            imgs = [im.to(dev).to(dt) for im in synth_images(batch, size, size, seed=1)]
        m.predict(imgs)
        torch.cuda.synchronize()
        print(f"{cfg}: {len(engine.Plan._TUNE_CACHE)} shape keys tuned so far", flush=True)
        del m, imgs
        torch.cuda.empty_cache()
    tiles = {engine.tile_key_str(k[:-1], k[-1]): int(v) for k, v in engine.Plan._TUNE_CACHE.items()}
    old = {}
    if args.merge and os.path.exists(args.out):
        old = json.load(open(args.out)).get("tiles", {})
    old.update(tiles)
    doc = {"device": torch.cuda.get_device_name(0), "note": "pinned conv tile ids per (shape, dtype); regenerate with tools/tune_tiles.py on an MI355X",
           "configs": args.configs, "tiles": dict(sorted(old.items())), "us_per_candidate": dict(sorted(engine.Plan._TUNE_TIMES.items()))}
    with open(args.out, "w") as fh:
        json.dump(doc, fh, indent=0, sort_keys=False)
    print(f"wrote {len(old)} entries to {args.out}")


if __name__ == "__main__":
    main()
