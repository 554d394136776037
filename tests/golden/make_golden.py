"""Generates tests/golden/*.npz|json by running the UNMODIFIED reference (read-only import from
/root/reference, torchvision stand-in from oracle/_tv_compat.py).  Build-container only.

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.reference_loader import load_reference  # noqa: E402
from yolort_amd.utils.synth import synth_images, synth_weights  # noqa: E402

yolort = load_reference()
from yolort.models import YOLOv5  # noqa: E402
from yolort.models.anchor_utils import AnchorGenerator  # noqa: E402
from yolort.models.box_head import _concat_pred_logits, _decode_pred_logits  # noqa: E402
from yolort.models.transform import YOLOTransform, scale_coords  # noqa: E402

LETTERBOX_SHAPES = [(1080, 810), (480, 640), (720, 1280), (1080, 1920), (375, 500), (427, 640), (333, 500),
                    (1281, 1279), (641, 480), (100, 37)]


def letterbox_golden():
    out = {"cases": []}
    for S in (640, 1280):
        for stride in (32, 64):
            t = YOLOTransform(S, S, size_divisible=stride)
            for (h, w) in LETTERBOX_SHAPES:
                img = synth_images(1, h, w, seed=h * 7 + w)[0]
                nt, _ = t([img])
                hh, ww = nt.image_sizes[0]
                canvas = tuple(nt.tensors.shape[-2:])
                x = nt.tensors[0]
                # locate top-left of the placed image via the reference's own rule is implicit; store checksums
                out["cases"].append({"S": S, "stride": stride, "hw": [h, w], "resized": [int(hh), int(ww)], "canvas": list(canvas),
                                     "sum": float(x.double().sum()), "sample": [float(v) for v in x[:, ::max(1, canvas[0] // 7), ::max(1, canvas[1] // 5)].flatten()[:64]]})
    # mixed batch + fixed shape
    for fixed in (None, (640, 640)):
        t = YOLOTransform(640, 640, fixed_shape=fixed)
        imgs = [synth_images(1, h, w, seed=h + w)[0] for (h, w) in [(1080, 810), (480, 640), (720, 1280)]]
        nt, _ = t(imgs)
        out["cases"].append({"mixed": True, "fixed": list(fixed) if fixed else None, "canvas": list(nt.tensors.shape),
                             "image_sizes": [list(map(int, s)) for s in nt.image_sizes], "sum": float(nt.tensors.double().sum())})
    t = YOLOTransform(640, 640, fixed_shape=(640, 640))
    nt, _ = t([synth_images(1, 427, 640, seed=3)[0]])
    rows = (nt.tensors[0, 0, :, 320] != 114 / 255).nonzero().flatten()
    out["fixed_427"] = {"first_row": int(rows[0]), "last_row": int(rows[-1])}
    out["fill"] = float(torch.tensor(114 / 255, dtype=torch.float32))
    b = scale_coords(torch.tensor([[100.0, 100.0, 200.0, 200.0]]), torch.tensor([640, 480]), (1080, 810))
    out["scale_coords"] = [float(v) for v in b.flatten()]
    b = scale_coords(torch.tensor([[10.5, 20.25, 300.75, 333.0]]), torch.tensor([384, 640]), (720, 1280))
    out["scale_coords2"] = [float(v) for v in b.flatten()]
    json.dump(out, open(os.path.join(HERE, "letterbox.json"), "w"))


def anchors_decode_golden():
    # reference test/test_models_anchor_utils.py:14-30 setup
    ag = AnchorGenerator([4], [[6, 14]])
    grids, shifts = ag([torch.rand(1, 3, 2, 2)])
    # SURVEY.md Appendix F setup
    ag2 = AnchorGenerator([8], [[10, 13, 16, 30, 33, 23]])
    ho = [torch.full((1, 3, 2, 3, 85), float(np.log(3.0)))]
    g2, s2 = ag2([torch.rand(1, 4, 2, 3)])
    pred = _concat_pred_logits(ho, g2, s2, torch.tensor([8.0]))
    boxes, scores = _decode_pred_logits(pred[0])
    # random head outputs through the reference decode (reference test pattern test/test_models.py:153-165)
    torch.manual_seed(7)
    ho3 = [torch.randn(2, 3, 4, 5, 85) * 2, torch.randn(2, 3, 2, 3, 85) * 2]
    ag3 = AnchorGenerator([8, 16], [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119]])
    g3, s3 = ag3([torch.rand(1, 1, 4, 5), torch.rand(1, 1, 2, 3)])
    pred3 = _concat_pred_logits(ho3, g3, s3, torch.tensor([8.0, 16.0]))
    np.savez(os.path.join(HERE, "anchors_decode.npz"), grids=grids[0].numpy(), shifts=shifts[0].numpy(),
             kat_pred=pred.numpy(), kat_boxes=boxes.numpy(), kat_scores=scores.numpy(),
             rnd_ho0=ho3[0].numpy(), rnd_ho1=ho3[1].numpy(), rnd_pred=pred3.numpy())


def e2e_golden(arch="yolov5_darknet_pan_n_r60", tag="n", sizes=((160, 120), (96, 160), (128, 128)), S=160, head_gain=1.0, thr=0.1):
    model = YOLOv5(arch=arch, size=(S, S), score_thresh=thr, nms_thresh=0.45)
    sd = synth_weights(model.state_dict(), arch, seed=0, head_gain=head_gain)
    model.load_state_dict(sd)
    model.eval()
    imgs = [synth_images(1, h, w, seed=11 + i)[0] for i, (h, w) in enumerate(sizes)]
    stages = {}
    model.model.backbone.register_forward_hook(lambda m, i, o: stages.__setitem__("features", o))
    model.model.head.register_forward_hook(lambda m, i, o: stages.__setitem__("head", o))
    model.transform.register_forward_hook(lambda m, i, o: stages.__setitem__("batch", o[0].tensors))
    with torch.no_grad():
        dets = model.predict(imgs)
    out = {"sizes": np.asarray(sizes), "S": S, "head_gain": head_gain, "thr": thr, "batch_shape": np.asarray(stages["batch"].shape)}
    out["batch_sum"] = float(stages["batch"].double().sum())
    for i, f in enumerate(stages["features"]):
        out[f"feat{i}"] = f.numpy().astype(np.float32)
    for i, h in enumerate(stages["head"]):
        out[f"head{i}"] = h.numpy().astype(np.float32)
    for i, d in enumerate(dets):
        out[f"det{i}_boxes"] = d["boxes"].numpy()
        out[f"det{i}_scores"] = d["scores"].numpy()
        out[f"det{i}_labels"] = d["labels"].numpy()
        print(tag, "image", i, "detections", len(d["scores"]), "labels", len(set(d["labels"].tolist())))
    np.savez_compressed(os.path.join(HERE, f"e2e_{tag}.npz"), **out)


if __name__ == "__main__":
    letterbox_golden()
    anchors_decode_golden()
    e2e_golden()
    print("golden written to", HERE)
