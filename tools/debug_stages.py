"""GPU debug helper: per-stage error of the HIP path vs the oracle (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import yolov5_oracle as O
from yolort_amd.models import YOLOv5
from workloads.synth import synth_images, synth_weights

dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "yolov5_darknet_pan_n_r60"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 160
m = YOLOv5(arch=arch, size=(S, S), score_thresh=0.3)
m.load_state_dict(synth_weights(m.state_dict(), arch, seed=0, head_gain=2.0))
m = m.to(dev).half().eval()
imgs = [synth_images(1, S, S * 3 // 4, seed=11)[0], synth_images(1, S // 2, S, seed=12)[0]]
dets = m.predict([im.to(dev) for im in imgs])
sd = {k: v.float().cpu() for k, v in m.state_dict().items()}

# per-conv reference via hooks on oracle: monkeypatch conv_bn_silu to record outputs by prefix
rec = {}
orig = O.conv_bn_silu
def rec_conv(x, sd_, p, stride=1, pad=None):
    y = orig(x, sd_, p, stride, pad)
    rec[p] = y
    return y
O.conv_bn_silu = rec_conv
if os.environ.get("EMU", "1") == "1":
    O.EMULATE.dtype = torch.float16
with torch.no_grad():
    ref, st = O.yolov5_forward([im.half().float() for im in imgs], sd, size=(S, S), score_thresh=0.3, return_stages=True)
e = next(iter(m.model._entries.values()))
xb = e.x.as_tensor().float().cpu()[..., :3].permute(0, 3, 1, 2)
print("letterbox err", (xb - st["batch"]).abs().max().item(), "pad ch", e.x.as_tensor()[..., 3].abs().max().item())
for i, v in enumerate(e.feats):
    got = v.as_tensor().float().cpu().permute(0, 3, 1, 2)
    r = st["features"][i]
    print(f"feat{i} shape {tuple(got.shape)} err {(got - r).abs().max().item():.4f} ref absmax {r.abs().max().item():.3f} got absmax {got.abs().max().item():.3f}")
for i, v in enumerate(e.logits):
    got = v.as_tensor().cpu().view(v.n, v.h, v.w, 3, 85).permute(0, 3, 1, 2, 4)
    print(f"head{i} err {(got - st['head'][i]).abs().max().item():.4f}")
for r, d in zip(ref, dets):
    print("dets", len(r["scores"]), len(d["scores"]), r["labels"][:10].tolist(), d["labels"][:10].tolist())
    print(r["boxes"][:3], d["boxes"][:3].cpu())
# run plan op by op and compare conv outputs by name where possible
plan = e.plan
print("ops", plan.num_ops)
