"""ultralytics checkpoint ingest without vendoring upstream code.  A synthetic
upstream-style checkpoint (pickled nn.Module tree under fake `models.yolo` / `models.common` classes,
upstream layer numbering) is written, the fake classes are removed again, and the loader must rebuild
the yolort-format state_dict from it (CPU only)."""
import sys
import types

import pytest
import torch
from torch import nn


def _fake_upstream_modules():
    common = types.ModuleType("models.common")
    yolo = types.ModuleType("models.yolo")
    pkg = types.ModuleType("models")

    class Conv(nn.Module):
        def __init__(self, c1, c2, k=1):
            super().__init__()
            self.conv = nn.Conv2d(c1, c2, k, bias=False)
            self.bn = nn.BatchNorm2d(c2)

    class Bottleneck(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.cv1, self.cv2 = Conv(c, c, 1), Conv(c, c, 3)

    class C3(nn.Module):
        def __init__(self, c1, c2, n):
            super().__init__()
            c_ = c2 // 2
            self.cv1, self.cv2, self.cv3 = Conv(c1, c_), Conv(c1, c_), Conv(2 * c_, c2)
            self.m = nn.Sequential(*[Bottleneck(c_) for _ in range(n)])

    class SPPF(nn.Module):
        def __init__(self, c1, c2):
            super().__init__()
            self.cv1, self.cv2 = Conv(c1, c1 // 2), Conv(c1 // 2 * 4, c2)

    class Concat(nn.Module):
        pass

    class Detect(nn.Module):
        def __init__(self, chs, nc, anchors, strides):
            super().__init__()
            self.m = nn.ModuleList(nn.Conv2d(c, 3 * (nc + 5), 1) for c in chs)
            s = torch.tensor(strides, dtype=torch.float32)
            self.register_buffer("anchors", torch.tensor(anchors, dtype=torch.float32).view(len(chs), -1, 2) / s.view(-1, 1, 1))
            self.stride = s

    class Model(nn.Module):
        pass

    for cls in (Conv, Bottleneck, C3, SPPF, Concat):
        cls.__module__, cls.__qualname__ = "models.common", cls.__name__
        setattr(common, cls.__name__, cls)
    for cls in (Detect, Model):
        cls.__module__, cls.__qualname__ = "models.yolo", cls.__name__
        setattr(yolo, cls.__name__, cls)
    pkg.common, pkg.yolo = common, yolo
    return {"models": pkg, "models.common": common, "models.yolo": yolo}, (Conv, C3, SPPF, Concat, Detect, Model)


def _write_fake_checkpoint(path, ref_sd, p6=False):
    """upstream yolov5n(6).yaml topology with the weights of `ref_sd` (yolort naming) placed by index"""
    from yolort_amd.models._checkpoint import _index_maps
    mods, (Conv, C3, SPPF, Concat, Detect, Model) = _fake_upstream_modules()
    sys.modules.update(mods)
    try:
        w = [16, 32, 64, 128, 192, 256] if p6 else [16, 32, 64, 128, 256]
        layers = [Conv(3, w[0], 6), Conv(w[0], w[1], 3), C3(w[1], w[1], 1), Conv(w[1], w[2], 3), C3(w[2], w[2], 2), Conv(w[2], w[3], 3), C3(w[3], w[3], 3)]
        if p6:
            layers += [Conv(w[3], w[4], 3), C3(w[4], w[4], 1), Conv(w[4], w[5], 3), C3(w[5], w[5], 1), SPPF(w[5], w[5]),
                       Conv(w[5], w[4]), nn.Upsample(scale_factor=2), Concat(), C3(2 * w[4], w[4], 1),
                       Conv(w[4], w[3]), nn.Upsample(scale_factor=2), Concat(), C3(2 * w[3], w[3], 1),
                       Conv(w[3], w[2]), nn.Upsample(scale_factor=2), Concat(), C3(2 * w[2], w[2], 1),
                       Conv(w[2], w[2], 3), Concat(), C3(2 * w[2], w[3], 1), Conv(w[3], w[3], 3), Concat(), C3(2 * w[3], w[4], 1),
                       Conv(w[4], w[4], 3), Concat(), C3(2 * w[4], w[5], 1)]
            chs, strides = [w[2], w[3], w[4], w[5]], [8, 16, 32, 64]
            anchors = [[19, 27, 44, 40, 38, 94], [96, 68, 86, 152, 180, 137], [140, 301, 303, 264, 238, 542], [436, 615, 739, 380, 925, 792]]
        else:
            layers += [Conv(w[3], w[4], 3), C3(w[4], w[4], 1), SPPF(w[4], w[4]),
                       Conv(w[4], w[3]), nn.Upsample(scale_factor=2), Concat(), C3(2 * w[3], w[3], 1),
                       Conv(w[3], w[2]), nn.Upsample(scale_factor=2), Concat(), C3(2 * w[2], w[2], 1),
                       Conv(w[2], w[2], 3), Concat(), C3(2 * w[2], w[3], 1), Conv(w[3], w[3], 3), Concat(), C3(2 * w[3], w[4], 1)]
            chs, strides = [w[2], w[3], w[4]], [8, 16, 32]
            anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
        layers.append(Detect(chs, 80, anchors, strides))
        model = Model()
        model.model = nn.Sequential(*layers)
        model.yaml = {"nc": 80, "depth_multiple": 0.33, "width_multiple": 0.25, "anchors": anchors}
        model.stride = torch.tensor(strides, dtype=torch.float32)
        index_map, head_idx = _index_maps(p6)
        sd = model.state_dict()
        for k in list(sd):
            parts = k.split(".")
            idx, rest = parts[1], ".".join(parts[2:])
            if idx == head_idx and parts[2] == "m":
                sd[k] = ref_sd["head.head." + ".".join(parts[3:])].clone()
            elif idx in index_map:
                sd[k] = ref_sd[index_map[idx] + "." + rest].clone()
        model.load_state_dict(sd)
        torch.save({"model": model, "epoch": -1, "ema": None}, path)
        return anchors, strides
    finally:
        for name in mods:
            sys.modules.pop(name, None)


@pytest.mark.parametrize("p6", [False, True])
def test_roundtrip_through_upstream_format(tmp_path, p6):
    from yolort_amd.models import yolo as Y
    from yolort_amd.models._checkpoint import load_from_ultralytics
    from workloads.synth import synth_state_dict
    arch = "yolov5_darknet_pan_n6_r60" if p6 else "yolov5_darknet_pan_n_r60"
    ref = Y.__dict__[arch]()
    ref_sd = synth_state_dict(ref.state_dict(), seed=3)
    path = str(tmp_path / "yolov5n_fake.pt")
    anchors, strides = _write_fake_checkpoint(path, ref_sd, p6)
    assert "models" not in sys.modules          # the loader must not need the upstream classes
    info = load_from_ultralytics(path)
    assert info["size"] == "n" and info["use_p6"] == p6 and info["num_classes"] == 80 and info["strides"] == strides
    assert info["anchor_grids"] == [[float(v) for v in row] for row in anchors]
    assert set(info["state_dict"]) == set(ref_sd)
    for k, v in ref_sd.items():
        got = info["state_dict"][k]
        if v.is_floating_point():
            assert got.dtype == torch.float16 and torch.equal(got.float(), v.half().float()), k   # fp16-rounded like the reference (:81)
        else:
            assert torch.equal(got, v), k
    model = Y.YOLO.load_from_yolov5(path, score_thresh=0.3)
    assert model.post_process.score_thresh == 0.3 and len(model.state_dict()) == len(ref_sd)
    from yolort_amd.models import YOLOv5
    wrapped = YOLOv5.load_from_yolov5(path, size=(320, 320), score_thresh=0.4)
    assert wrapped.transform.min_size == 320 and wrapped.model.post_process.score_thresh == 0.4


def test_loader_refuses_foreign_code(tmp_path):
    """a pickle that names an arbitrary callable must not execute it: it becomes an inert stub"""
    import pickle

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /tmp/ymi_pwned",))

    path = str(tmp_path / "evil.pt")
    torch.save({"model": Evil()}, path)
    from yolort_amd.models._checkpoint import load_from_ultralytics
    import os
    if os.path.exists("/tmp/ymi_pwned"):
        os.remove("/tmp/ymi_pwned")
    with pytest.raises(Exception):
        load_from_ultralytics(path)
    assert not os.path.exists("/tmp/ymi_pwned")

    class Evil2:
        def __reduce__(self):
            return (eval, ("open('/tmp/ymi_pwned', 'w').write('x')",))

    torch.save({"model": Evil2()}, path)
    with pytest.raises(Exception):
        load_from_ultralytics(path)
    assert not os.path.exists("/tmp/ymi_pwned")


def _global_call_pickle(module, name, *args):
    """protocol-4 pickle of `module.name(*args)` (string args), built opcode by opcode -- `name` may be dotted"""
    import pickle as P

    def s(v):
        b = v.encode()
        return P.SHORT_BINUNICODE + bytes([len(b)]) + b

    out = P.PROTO + b"\x04" + s(module) + s(name) + P.STACK_GLOBAL + P.MARK
    for a in args:
        out += s(a)
    return out + P.TUPLE + P.REDUCE + P.STOP


@pytest.mark.parametrize("module,name", [
    ("torch._utils", "_import_dotted_name"),                 # resolves any dotted path when called: os.system one call away
    ("torch._utils", "traceback.linecache.os.getcwd"),       # protocol-4 dotted name: attribute traversal out of an allowed module
    ("collections", "_sys.getrecursionlimit"),
    ("collections", "_sys"),
    ("numpy", "memmap"),                                     # a real type whose constructor touches the file system
    ("torch", "load"), ("torch.hub", "load"), ("os", "system"), ("builtins", "eval"), ("builtins", "getattr"),
])
def test_unpickler_resolves_nothing_outside_the_allow_list(module, name):
    """module-level trust let crafted pickles reach real callables.  Every one of these must come back as
    an inert stub class, both from find_class and when the pickle calls it."""
    import io

    from yolort_amd.models._checkpoint import _StubUnpickler, _UpstreamStub
    cls = _StubUnpickler(io.BytesIO(b"")).find_class(module, name)
    assert isinstance(cls, type) and issubclass(cls, _UpstreamStub), (module, name, cls)
    obj = _StubUnpickler(io.BytesIO(_global_call_pickle(module, name, "os.getcwd"))).load()
    assert isinstance(obj, _UpstreamStub)


def test_unpickler_still_resolves_the_data_constructors():
    import collections
    import io

    from yolort_amd.models._checkpoint import _StubUnpickler
    u = _StubUnpickler(io.BytesIO(b""))
    assert u.find_class("collections", "OrderedDict") is collections.OrderedDict
    assert u.find_class("torch._utils", "_rebuild_tensor_v2") is torch._utils._rebuild_tensor_v2
    assert u.find_class("torch", "float16") is torch.float16 and u.find_class("torch", "HalfStorage") is torch.HalfStorage
    assert u.find_class("torch.nn.modules.conv", "Conv2d") is nn.Conv2d
    assert u.find_class("builtins", "set") is set


# ---- pinned to the REFERENCE's converter (round 5) --------------------------------------------------------------------------------------
# tests/golden/make_checkpoint_golden.py built ultralytics-format checkpoints with the reference's vendored upstream classes, ran the UNMODIFIED reference's
# `load_from_ultralytics` (yolort/models/_checkpoint.py:16-94, `.half()` at :81) on them and committed the ordered key list, a sha256 per tensor and the metadata
# (tests/golden/ckpt_golden.json) plus the yolov5n checkpoint file.  This repo's converter unpickles with inert stubs instead of the upstream classes and must
# reproduce that output bit for bit.
import hashlib
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tensor_sha(t):
    h = hashlib.sha256()
    h.update(str(t.dtype).encode())
    h.update(str(tuple(t.shape)).encode())
    h.update(t.detach().contiguous().cpu().reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    return h.hexdigest()


def _assert_equals_reference_record(info, rec):
    sd = info["state_dict"]
    assert list(sd.keys()) == rec["keys"]                                   # same tensors under the same names, in the reference's order
    assert sorted({str(v.dtype) for v in sd.values()}) == rec["dtypes"]     # fp16-rounded weights, int64 counters
    bad = [k for k, v in sd.items() if _tensor_sha(v) != rec["sha256"][k]]
    assert not bad, f"{len(bad)} tensors differ from the reference's conversion, first: {bad[:5]}"
    assert int(info["num_classes"]) == rec["num_classes"] and info["size"] == rec["size"] and bool(info["use_p6"]) == rec["use_p6"]
    assert float(info["depth_multiple"]) == rec["depth_multiple"] and float(info["width_multiple"]) == rec["width_multiple"]
    assert [float(s) for s in info["strides"]] == rec["strides"]
    assert [[float(v) for v in row] for row in info["anchor_grids"]] == rec["anchor_grids"]


def test_converter_reproduces_the_reference_conversion_of_an_upstream_checkpoint_bit_for_bit():
    """the committed yolov5n checkpoint (pickled by the reference's vendored upstream classes) through this repo's stub unpickler == the reference's own
    load_from_ultralytics output recorded in ckpt_golden.json: 348 tensors, names, order, dtypes, every byte"""
    from yolort_amd.models._checkpoint import load_from_ultralytics
    with open(os.path.join(GOLD, "ckpt_golden.json")) as f:
        rec = json.load(f)["archs"]["n"]
    path = os.path.join(GOLD, "yolov5n_upstream_format.pt")
    assert os.path.getsize(path) == rec["checkpoint_bytes"]
    info = load_from_ultralytics(path)
    assert len(info["state_dict"]) == 348
    _assert_equals_reference_record(info, rec)
    # ... and the result loads into the model it describes (YOLO.load_from_yolov5's last step, yolo.py:222)
    from yolort_amd.models import yolo
    model = yolo.yolov5_darknet_pan_n_r60(num_classes=info["num_classes"])
    missing, unexpected = model.load_state_dict(info["state_dict"], strict=True)
    assert not missing and not unexpected


@pytest.mark.parametrize("tag", ["n", "s", "m", "l", "n6", "s_r40", "s_r31"])   # (s_r40 / s_r31, round 5: upstream-format checkpoints of the legacy releases -- Focus, SPP in the backbone, BottleneckCSP)
def test_converter_against_the_live_reference_every_size(tag, tmp_path):
    """build container only: the checkpoint of each size is rebuilt with the reference's vendored upstream classes (same seeds as the committed record), converted by
    the UNMODIFIED reference and by this repo, and the two state_dicts are compared tensor by tensor -- and with the committed hashes (s / m / l / n6 files are 7-94 MB
    and are not committed)"""
    from oracle.reference_loader import reference_available
    if not reference_available():
        pytest.skip("needs /root/reference (build container)")
    os.environ.setdefault("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", "1")   # the reference calls torch.load without weights_only (torch >= 2.6 defaults to True)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_checkpoint_golden", os.path.join(GOLD, "make_checkpoint_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    from oracle.reference_loader import load_reference
    load_reference()
    from yolort.models._checkpoint import load_from_ultralytics as ref_load
    from yolort_amd.models._checkpoint import load_from_ultralytics
    path = str(tmp_path / f"yolov5{tag}.pt")
    mk.build_upstream_checkpoint(tag, path)
    version = mk.VERSION.get(tag, "r6.0")
    ref, mine = ref_load(path, version=version), load_from_ultralytics(path, version=version)
    a, b = ref["state_dict"], mine["state_dict"]
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    with open(os.path.join(GOLD, "ckpt_golden.json")) as f:
        rec = json.load(f)["archs"][tag]
    _assert_equals_reference_record(mine, rec)
    if version != "r6.0":   # ... and the converted weights load into the model of that release (YOLO.load_from_yolov5, reference yolo.py:186-223)
        from yolort_amd.models import yolo
        model = yolo.YOLO.load_from_yolov5(path, version=version)
        assert type(model.backbone.body["0"]).__name__ == "Focus" and len(model.state_dict()) == len(b)
        assert all(torch.equal(model.state_dict()[k].half(), b[k]) for k in b if b[k].is_floating_point())
