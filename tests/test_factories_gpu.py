"""Every model factory of the reference (yolort/models/yolo.py:292-834: n / s / m / l / x, their P6 variants, the r3.1 / r4.0 releases of s / m / l) END TO END in fp32
mode against the oracle, at a small canvas.  What is asserted per factory:
  * the pyramid features the conv stack hands to the head agree with the oracle's fp32 forward within 5e-3 of each feature's range (exact fp32 products, other summation
    order; measured 2e-5 .. 4e-4, and 1e-3 .. 2e-3 on the deepest level of x / x6 / l-r3.1, whose activations reach 10^2 .. 10^3 on this workload: profiles/r06m_*) -- a deterministic check of the whole backbone + PAN, independent of where the synthetic network's scores fall;
  * the detections pair with the oracle's (same label, IoU >= 0.99, |dscore| <= 1e-3) for >= 90 % of those not within 2e-3 of the score threshold / of the top-300
    cut (the seeded synthetic networks either saturate the cut or sit under the threshold -- some factories produce no detection at all on this workload, then the HIP
    path must produce (next to) none as well; the 18 goldens of tests/test_golden_gpu.py carry the exact-pairing claim on conditioned workloads).
The per-launch parity of the same factories' 16-bit plans is tests/test_parity_gpu.py::test_every_conv_launch_of_the_plan_vs_oracle_layer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ARCHS = ["yolov5_darknet_pan_n_r60", "yolov5_darknet_pan_s_r60", "yolov5_darknet_pan_m_r60", "yolov5_darknet_pan_l_r60", "yolov5_darknet_pan_x_r60",
         "yolov5_darknet_pan_n6_r60", "yolov5_darknet_pan_s6_r60", "yolov5_darknet_pan_m6_r60", "yolov5_darknet_pan_l6_r60", "yolov5_darknet_pan_x6_r60",
         "yolov5_darknet_pan_s_r31", "yolov5_darknet_pan_m_r31", "yolov5_darknet_pan_l_r31", "yolov5_darknet_pan_s_r40", "yolov5_darknet_pan_m_r40", "yolov5_darknet_pan_l_r40"]


@pytest.fixture(scope="module")
def dev():
    from yolort_amd import _lib
    _lib.load(require_gpu=True)
    return torch.device("cuda:0")


def _iou(a, b):
    x1, y1 = np.maximum(a[0], b[:, 0]), np.maximum(a[1], b[:, 1])
    x2, y2 = np.minimum(a[2], b[:, 2]), np.minimum(a[3], b[:, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter + 1e-12)


@pytest.mark.parametrize("arch", ARCHS)
def test_factory_fp32_mode_end_to_end_vs_oracle(dev, arch):
    from oracle import yolov5_oracle as O
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    thr, S = 0.1, 320
    kw = dict(size_divisible=64) if arch.endswith("6_r60") else {}
    m = YOLOv5(arch=arch, size=(S, S), score_thresh=thr, **kw)
    sd = synth_weights(m.state_dict(), arch, seed=0, head_gain=0.8)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.set_compute_dtype(torch.float32)
    imgs_cpu = [synth_images(1, 256, 320, seed=31)[0], synth_images(1, 320, 192, seed=32)[0]]
    dets = m.predict([im.to(dev) for im in imgs_cpu])
    torch.cuda.synchronize()
    e = next(iter(m.model._entries.values()))
    assert e.plan.fp32
    sdf = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        batch, _ = O.letterbox(imgs_cpu, S, S, kw.get("size_divisible", 32))
        feats = O.backbone(batch, sdf, "model.backbone")
        ref = O.yolov5_forward(imgs_cpu, sdf, size=(S, S), size_divisible=kw.get("size_divisible", 32), score_thresh=thr)
    assert len(feats) == len(e.feats) == (4 if kw else 3)
    worst = 0.0
    for f, v in zip(feats, e.feats):
        got = v.as_tensor().float().cpu().permute(0, 3, 1, 2)
        assert got.shape == f.shape, (got.shape, f.shape)
        err, scale = float((got - f).abs().max()), float(f.abs().max())
        worst = max(worst, err / scale)
        assert err <= 5e-3 * scale, f"{arch}: feature {tuple(f.shape)} |hip - oracle| {err:.3g} vs range {scale:.3g}"
    n_ref = n_got = n_pair = n_clear = 0
    for r, d in zip(ref, dets):
        rb, rs, rl = (r[k].detach().cpu().numpy() for k in ("boxes", "scores", "labels"))
        gb, gs, gl = (d[k].detach().cpu().numpy() for k in ("boxes", "scores", "labels"))
        n_ref += len(rs)
        n_got += len(gs)
        cut = max(thr, float(rs[-1]) if len(rs) >= 300 else 0.0)   # (scores come sorted: the last of 300 is the top-k cut)
        for i in range(len(rs)):
            if rs[i] < cut + 2e-3:
                continue
            n_clear += 1
            cand = np.where(gl == rl[i])[0]
            if len(cand) and np.any((_iou(rb[i], gb[cand]) >= 0.99) & (np.abs(gs[cand] - rs[i]) <= 1e-3)):
                n_pair += 1
    print(f"{arch}: worst feature error {worst:.2e} of range; {n_pair} of {n_clear} clear reference detections paired ({n_ref} reference / {n_got} HIP detections in all)")
    assert abs(n_got - n_ref) <= max(3, n_ref // 20), (n_got, n_ref)
    assert n_pair >= 0.9 * n_clear, (n_pair, n_clear)
