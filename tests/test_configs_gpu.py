"""BASELINE.json configs[2] (C3) and configs[4] (C5) AT SPEC on the MI355X (both had only run reduced).

  C3  yolov5m bf16, batch 64, size (1280, 1280), the 8 cycled image shapes of SURVEY.md 8d (they hit the letterbox rounding
      traps of App. B): per-image bilinear gather + common canvas for the whole 64-image list, then the bf16 conv stack.
  C5  yolov5l6 fp16, batch 8, 1280x1280, four pyramid levels (102 000 anchors / image), >= 5 000 candidates per image and
      300 kept per image: the class-aware NMS / sort / top-K stress.
The oracle (CPU fp32) is run on a bounded subset of each batch -- images are independent given the common canvas.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

C3_SHAPES = [(1080, 1920), (720, 1280), (1920, 1080), (1080, 810), (960, 1280), (1281, 1279), (641, 480), (375, 500)]


@pytest.fixture(scope="module")
def dev():
    from yolort_amd import _lib
    _lib.load(require_gpu=True)
    return torch.device("cuda:0")


def _np(d):
    return {k: (v.detach().float().cpu().numpy() if k != "labels" else v.detach().cpu().numpy()) for k, v in d.items()}


def test_config3_yolov5m_bf16_bs64_dynamic_1280_at_spec(dev):
    from oracle import yolov5_oracle as O
    from test_e2e_gpu import match_fraction
    from yolort_amd.models import yolov5m
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_m_r60"
    m = yolov5m(size=(1280, 1280), score_thresh=0.3)
    sd = synth_weights(m.state_dict(), arch, seed=0, head_gain=2.0)
    m.load_state_dict(sd)
    m = m.to(dev).to(torch.bfloat16).eval()
    imgs_cpu = [synth_images(1, *C3_SHAPES[i % 8], seed=100 + i)[0] for i in range(64)]
    imgs = [im.to(dev).to(torch.bfloat16) for im in imgs_cpu]
    dets = m.predict(imgs)
    torch.cuda.synchronize()
    e = next(iter(m.model._entries.values()))
    assert (e.x.n, e.x.h, e.x.w) == (64, 1280, 1280)
    # (1) the letterboxed 64-image batch against the oracle's letterbox (bf16 inputs, fp32 lerp, one bf16 rounding)
    with torch.no_grad():
        ref_batch, ref_sizes = O.letterbox([im.to(torch.bfloat16).float() for im in imgs_cpu], 1280, 1280, 32)
    assert tuple(ref_batch.shape) == (64, 3, 1280, 1280)
    got = e.x.as_tensor()[..., :3].float().cpu().permute(0, 3, 1, 2)
    want = ref_batch.to(torch.bfloat16).float()
    diff = (got - want).abs()
    # identical up to the last bf16 bit (the fp32 lerp may land on either side of a rounding boundary)
    assert float(diff.max()) <= 2 ** -8 + 1e-6, float(diff.max())
    assert float((diff > 0).float().mean()) < 2e-3, float((diff > 0).float().mean())
    assert bool((e.x.as_tensor()[..., 3] == 0).all())
    # (2) detections of one image per trap shape against the oracle (fp32) and its bf16-storage emulation
    pick = [0, 2, 5, 6]                       # 1080x1920, 1920x1080, 1281x1279, 641x480: their common canvas is again 1280x1280
    sdf = {k: v.float() for k, v in sd.items()}
    sub = [imgs_cpu[i] for i in pick]
    with torch.no_grad():
        ref = O.yolov5_forward(sub, sdf, size=(1280, 1280), score_thresh=0.3)
        O.EMULATE.dtype = torch.bfloat16
        try:
            emu = O.yolov5_forward([im.to(torch.bfloat16).float() for im in sub], sdf, size=(1280, 1280), score_thresh=0.3)
        finally:
            O.EMULATE.dtype = None
    fr, fe = [], []
    for r, em, i in zip(ref, emu, pick):
        f, miou, _ = match_fraction(_np(r), _np(dets[i]), iou_thr=0.5, score_tol=0.15, margin=0.1, thr=0.3)
        f_e, _, _ = match_fraction(_np(r), _np(em), iou_thr=0.5, score_tol=0.15, margin=0.1, thr=0.3)
        fr.append(f)
        fe.append(f_e)
        assert len(r["scores"]) > 10
    print(f"C3 at spec: matched (IoU>=0.5) HIP vs fp32 oracle {np.round(fr, 3)}, bf16-emulating oracle vs fp32 oracle {np.round(fe, 3)}")
    assert np.mean(fr) >= np.mean(fe) - 0.05 and all(a >= b - 0.2 for a, b in zip(fr, fe)), (fr, fe)
    assert np.mean(fr) >= 0.6
    # every image of the batch produced detections in ORIGINAL image coordinates
    for d in dets:
        assert len(d["scores"]) > 0 and bool(torch.isfinite(d["boxes"]).all())


def test_config5_yolov5l6_fp16_bs8_1280_k300_at_spec(dev):
    from oracle import yolov5_oracle as O
    from test_e2e_gpu import match_fraction
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    arch = "yolov5_darknet_pan_l6_r60"
    thr = 0.25
    m = YOLOv5(arch=arch, size=(1280, 1280), size_divisible=64, score_thresh=thr, nms_thresh=0.45, detections_per_img=300)
    sd = synth_weights(m.state_dict(), arch, seed=0, head_gain=3.0)   # >= 5 k candidates on every one of the 8 images (6 k ... 218 k)
    m.load_state_dict(sd)
    m = m.to(dev).half().eval()
    imgs_cpu = list(synth_images(8, 1280, 1280, seed=1))
    imgs = [im.to(dev).half() for im in imgs_cpu]
    dets = m.predict(imgs)                                   # fused head + decode, score-prefix selection
    e = next(iter(m.model._entries.values()))
    assert len(e.feats) == 4 and e.post.total_anchors == 102000
    n_cand_total = int(e.post.status[0].item())
    # the unfused form keeps the fp32 logits: same detections bit for bit, and the logits feed the oracle's post-process
    m.model.fuse_head_decode = False
    dets_u = m.predict(imgs)
    e = next(iter(m.model._entries.values()))
    for a, b in zip(dets, dets_u):
        for k in ("boxes", "scores", "labels"):
            assert torch.equal(a[k], b[k]), f"fused / unfused head disagree on {k} at C5"
    logits = [v.as_tensor().cpu().view(v.n, v.h, v.w, 3, 85).permute(0, 3, 1, 2, 4).contiguous() for v in e.logits]
    strides, anchors = O.anchors_for(4)
    with torch.no_grad():
        pred = O.decode(logits, strides, anchors)
        n_cand = [int(((pred[i, :, 5:] * pred[i, :, 4:5]) > thr).sum()) for i in range(8)]
        ref_post = O.postprocess(pred, thr, 0.45, 300)
    print(f"C5 at spec: candidates per image {n_cand} (HIP processed {n_cand_total} records after prefix selection), kept {[len(d['scores']) for d in dets]}")
    assert min(n_cand) >= 5000, n_cand                       # >= 5 k candidates / image ...
    assert all(len(d["scores"]) == 300 for d in dets)        # ... and 300 kept
    # (1) sort + class-aware NMS + top-K at this crowding, given identical logits: the same 300 detections.  On the most crowded image the scores saturate (0.99998...)
    # and the hardware's exp2 / rcp and torch's CPU sigmoid round a few of them one unit in the last place apart, which permutes detections of (almost) equal score:
    # the order is compared up to such ties (round 4: the re-tuned table changed the logits and with them which scores tie; profiles/r04p_c5_tie_case.txt)
    for i, (r, d) in enumerate(zip(ref_post, dets)):
        hl, hs, hb = d["labels"].cpu().numpy(), d["scores"].cpu().numpy(), d["boxes"].cpu().numpy()
        rl, rs, rb = r["labels"].numpy(), r["scores"].numpy(), r["boxes"].numpy()
        np.testing.assert_allclose(hs, rs, rtol=1e-5, atol=1e-6, err_msg=f"image {i}")       # the score SEQUENCE agrees (ties permute equal values)
        used = np.zeros(len(rl), bool)
        for j in range(len(hl)):   # every HIP detection is an oracle detection (label, box, score), at a position whose score is within two units in the last place
            c = np.where((rl == hl[j]) & ~used & (np.abs(rs - hs[j]) <= 2.5e-7) & (np.abs(rb - hb[j]).max(axis=1) <= 2e-3))[0]
            assert len(c) > 0, f"image {i}: detection {j} (label {hl[j]}, score {hs[j]}) has no counterpart"
            k = c[np.argmin(np.abs(c - j))]
            used[k] = True
            assert k == j or abs(float(rs[k]) - float(rs[j])) <= 2.5e-7, f"image {i}: detection {j} sits at {k} in the oracle's order, beyond a score tie"
        assert used.all()
    # (2) end to end against the fp32 oracle on two of the eight images (fp16-storage yardstick as in test_parity_gpu)
    sdf = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        ref = O.yolov5_forward(imgs_cpu[:2], sdf, size=(1280, 1280), size_divisible=64, score_thresh=thr)
        O.EMULATE.dtype = torch.float16
        try:
            emu = O.yolov5_forward([im.half().float() for im in imgs_cpu[:2]], sdf, size=(1280, 1280), size_divisible=64, score_thresh=thr)
        finally:
            O.EMULATE.dtype = None
    for r, em, d in zip(ref, emu, dets[:2]):
        f, miou, _ = match_fraction(_np(r), _np(d), iou_thr=0.9, score_tol=0.05, margin=0.03, thr=thr)
        f_e, _, _ = match_fraction(_np(r), _np(em), iou_thr=0.9, score_tol=0.05, margin=0.03, thr=thr)
        print(f"   matched (IoU>=0.9) HIP vs fp32 oracle {f:.3f}, fp16-emulating oracle vs fp32 oracle {f_e:.3f}, median IoU {miou:.4f}")
        assert f >= f_e - 0.15 and miou >= 0.93, (f, f_e, miou)
