"""Pins yolort_amd.models._checkpoint to the REFERENCE's own converter.  Build container only (imports /root/reference read-only).

    TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 python tests/golden/make_checkpoint_golden.py            -> tests/golden/ckpt_golden.json, tests/golden/yolov5n_upstream_format.pt

For yolov5 n / s / m / l and the P6 model n6: an ultralytics-FORMAT checkpoint is built from the reference's vendored upstream classes
(yolort/v5/models/yolo.py `Model(cfg=<yaml>)`, imported as top-level `models.yolo` under `yolort.v5.helper.add_yolov5_context()`, so the pickle names
`models.yolo.Model`, `models.common.Conv`, ... exactly like a file written by ultralytics/yolov5's train.py: {"model": model.half(), "ema": None, ...}),
seeded so that every tensor is distinct, and run through the reference's `load_from_ultralytics` (yolort/models/_checkpoint.py:16-94: index maps :53-64,
CheckpointConverter.updating :176-215, `.half().state_dict()` :81).  Committed: the ordered key list, a sha256 per tensor (dtype + shape + bytes), the
returned metadata -- and the yolov5n checkpoint file itself (3.9 MB), so that the converter test runs where the reference is absent.
(`TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1`: the reference calls torch.load without `weights_only`, which torch >= 2.6 defaults to True; the variable restores the
behaviour the reference was written for without touching its source.)
"""
import copy
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", "1")
import torch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
V5 = "/root/reference/yolort/v5/models"
ARCHS = {"n": (f"{V5}/yolov5n.yaml", 11), "s": (f"{V5}/yolov5s.yaml", 12), "m": (f"{V5}/yolov5m.yaml", 13), "l": (f"{V5}/yolov5l.yaml", 14), "n6": (f"{V5}/hub/yolov5n6.yaml", 15),
         "s_r40": ("legacy:C3", 16), "s_r31": ("legacy:BottleneckCSP", 17)}   # round 5: the legacy releases (the reference vendors no yaml for them: LEGACY_YAML below)
VERSION = {"s_r40": "r4.0", "s_r31": "r3.1"}

# ultralytics/yolov5's model description of the r3.1 / r4.0 yolov5s (Focus stem, SPP inside the backbone, the neck's first block as backbone layer 9), restated in the upstream
# yaml format; BLOCK = C3 (r4.0) or BottleneckCSP (r3.1).  The layer indices are the ones the reference's index maps assume (_checkpoint.py:53-64).
LEGACY_YAML = """
nc: 80
depth_multiple: 0.33
width_multiple: 0.50
anchors:
  - [10,13, 16,30, 33,23]
  - [30,61, 62,45, 59,119]
  - [116,90, 156,198, 373,326]
backbone:
  [[-1, 1, Focus, [64, 3]],
   [-1, 1, Conv, [128, 3, 2]],
   [-1, 3, BLOCK, [128]],
   [-1, 1, Conv, [256, 3, 2]],
   [-1, 9, BLOCK, [256]],
   [-1, 1, Conv, [512, 3, 2]],
   [-1, 9, BLOCK, [512]],
   [-1, 1, Conv, [1024, 3, 2]],
   [-1, 1, SPP, [1024, [5, 9, 13]]],
   [-1, 3, BLOCK, [1024, False]],
  ]
head:
  [[-1, 1, Conv, [512, 1, 1]],
   [-1, 1, nn.Upsample, [None, 2, 'nearest']],
   [[-1, 6], 1, Concat, [1]],
   [-1, 3, BLOCK, [512, False]],
   [-1, 1, Conv, [256, 1, 1]],
   [-1, 1, nn.Upsample, [None, 2, 'nearest']],
   [[-1, 4], 1, Concat, [1]],
   [-1, 3, BLOCK, [256, False]],
   [-1, 1, Conv, [256, 3, 2]],
   [[-1, 14], 1, Concat, [1]],
   [-1, 3, BLOCK, [512, False]],
   [-1, 1, Conv, [512, 3, 2]],
   [[-1, 10], 1, Concat, [1]],
   [-1, 3, BLOCK, [1024, False]],
   [[17, 20, 23], 1, Detect, [nc, anchors]],
  ]
"""


def tensor_sha(t: torch.Tensor) -> str:
    h = hashlib.sha256()
    h.update(str(t.dtype).encode())
    h.update(str(tuple(t.shape)).encode())
    h.update(t.detach().contiguous().cpu().reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    return h.hexdigest()


def build_upstream_checkpoint(tag: str, path: str) -> None:
    """an ultralytics-format checkpoint of architecture `tag` with seeded, all-distinct tensors (the reference's vendored upstream classes do the pickling)"""
    from yolort.v5.helper import add_yolov5_context

    cfg, seed = ARCHS[tag]
    if cfg.startswith("legacy:"):
        import tempfile
        tmp = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
        tmp.write(LEGACY_YAML.replace("BLOCK", cfg.split(":")[1]))
        tmp.close()
        cfg = tmp.name
    with add_yolov5_context():
        import models.yolo as upstream   # yolort/v5/models/yolo.py under its upstream name

        torch.manual_seed(seed)
        m = upstream.Model(cfg=cfg, ch=3, nc=80)
        g = torch.Generator().manual_seed(1000 + seed)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):   # defaults (ones / zeros) would make every BatchNorm tensor of a width identical: a swapped layer would go unseen
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
                mod.num_batches_tracked.fill_(int(torch.randint(1, 1000, (1,), generator=g)))
        ckpt = {"epoch": -1, "best_fitness": None, "model": copy.deepcopy(m).half(), "ema": None, "updates": None, "optimizer": None, "wandb_id": None}
        torch.save(ckpt, path)


def reference_conversion(path: str, version: str = "r6.0") -> dict:
    from yolort.models._checkpoint import load_from_ultralytics

    info = load_from_ultralytics(path, version=version)
    sd = info["state_dict"]
    return {"num_classes": int(info["num_classes"]), "depth_multiple": float(info["depth_multiple"]), "width_multiple": float(info["width_multiple"]),
            "strides": [float(s) for s in torch.as_tensor(info["strides"]).tolist()], "anchor_grids": [[float(v) for v in row] for row in info["anchor_grids"]],
            "use_p6": bool(info["use_p6"]), "size": info["size"], "keys": list(sd.keys()), "sha256": {k: tensor_sha(v) for k, v in sd.items()},
            "dtypes": sorted({str(v.dtype) for v in sd.values()})}


def main():
    from oracle.reference_loader import load_reference

    load_reference()
    import tempfile

    out = {"what": "outputs of the UNMODIFIED reference's load_from_ultralytics (yolort/models/_checkpoint.py:16-94) on seeded ultralytics-format checkpoints built "
                   "with its vendored upstream classes (tests/golden/make_checkpoint_golden.py)", "archs": {}}
    with tempfile.TemporaryDirectory() as td:
        for tag in ARCHS:
            path = os.path.join(GOLD, "yolov5n_upstream_format.pt") if tag == "n" else os.path.join(td, f"yolov5{tag}.pt")
            build_upstream_checkpoint(tag, path)
            rec = reference_conversion(path, VERSION.get(tag, "r6.0"))
            rec["version"] = VERSION.get(tag, "r6.0")
            rec["checkpoint_bytes"] = os.path.getsize(path)
            out["archs"][tag] = rec
            print(tag, len(rec["keys"]), "tensors", rec["checkpoint_bytes"], "bytes", rec["size"], rec["use_p6"], rec["strides"], flush=True)
    with open(os.path.join(GOLD, "ckpt_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
