"""TOOLS ONLY (never on the product path): what the vendor GEMM library gives for the M x N x K of every convolution of a plan.

VERDICT r3 item 6: the MFMA-bound shapes of the conv stack (the 3x3 layers at C3 / C5 sizes, the strided 3x3 layers of C2) reach 0.5-0.9 PFLOP/s with
the hand-written kernels; this script times a plain fp16 / bf16 GEMM of the same M = N_img * Ho * Wo, N = cout, K = cin * kh * kw through torch.mm
(hipBLASLt / rocBLAS behind it) on the same GPU, as a yardstick for what the chip delivers on that shape with a tuned library main loop and NO
convolution overheads (no im2col gather, no halo, no epilogue beyond the store).  Output: one line per distinct shape with the library's time next to
the plan's own per-op time (ymi_plan_profile).

usage: python tools/gemm_yardstick.py [c2|c3|c5] > profiles/r04_gemm_yardstick_<cfg>.txt
"""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    c = bench.CONFIGS[cfg]
    from yolort_amd.models import YOLOv5
    from workloads.synth import synth_images, synth_weights
    dev = torch.device("cuda:0")
    dt = torch.float16 if c["dtype"] == "fp16" else torch.bfloat16
    kw = dict(size_divisible=64) if c["arch"].endswith("6_r60") else {}
    m = YOLOv5(arch=c["arch"], size=(c["size"], c["size"]), score_thresh=c["score_thresh"], **kw)
    m.load_state_dict(synth_weights(m.state_dict(), c["arch"], seed=0, head_gain=c["head_gain"]))
    m = m.to(dev).to(dt).eval()
    m.model.stem_from_planar = False
    imgs = [im.to(dev).to(dt) for im in synth_images(c["batch"], c["size"], c["size"], seed=1)]
    for _ in range(2):
        m(imgs)
    torch.cuda.synchronize()
    e = next(iter(m.model._entries.values()))
    prof = e.plan.profile(iters=10)
    seen = {}
    print(f"# {cfg}: {c['arch']} {c['dtype']} bs {c['batch']} {c['size']}^2 -- per conv op: this repo's launch (ymi_plan_profile, us) vs torch.mm of the same M x N x K (hipBLASLt/rocBLAS, us)")
    print("# op, shape, M, N, K, ours_us, ours_TF, lib_gemm_us, lib_TF, ours/lib")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for name, ms, meta in prof:
        if meta.get("kind") != "conv" or "->" not in meta.get("shape", ""):
            continue
        mt = re.match(r"(\d+)->(\d+) k(\d+)x(\d+) s(\d+) (\d+)x(\d+)->(\d+)x(\d+)", meta["shape"])
        if not mt:
            print(f"{name}, {meta['shape']}, -, -, -, {ms * 1e3:.1f}, {meta['flops'] / ms / 1e9:.0f}, -, -, -")
            continue
        cin, cout, kh, kw_, s, h, w, ho, wo = [int(v) for v in mt.groups()]
        M, N, K = c["batch"] * ho * wo, cout, cin * kh * kw_
        key = (M, N, K)
        if key not in seen:
            a = torch.randn(M, K, device=dev, dtype=dt)
            b = torch.randn(K, N, device=dev, dtype=dt)
            out = torch.empty(M, N, device=dev, dtype=dt)
            for _ in range(3):
                torch.mm(a, b, out=out)
            torch.cuda.synchronize()
            # 20 calls captured in one graph: torch.mm's host-side launch cost (~19 us per call: the first version of this tool measured THAT for every small shape,
            # profiles/r04a_gemm_yardstick_c2_hostbound.txt) stays out of the measurement
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(20):
                        torch.mm(a, b, out=out)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                ev[0].record()
                g.replay()
                ev[1].record()
                ev[1].synchronize()
                best = min(best, ev[0].elapsed_time(ev[1]) / 20)
            del g, out
            seen[key] = best
            del a, b
        lib = seen[key]
        fl = 2.0 * M * N * K
        print(f"{name}, {meta['shape']}, {M}, {N}, {K}, {ms * 1e3:.1f}, {fl / ms / 1e9:.0f}, {lib * 1e3:.1f}, {fl / lib / 1e9:.0f}, {ms / lib:.2f}")


if __name__ == "__main__":
    main()
