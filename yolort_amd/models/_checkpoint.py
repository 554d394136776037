"""Ingest of ultralytics/yolov5 checkpoints (r3.1 / r4.0 / r6.0) (reference yolort/models/_checkpoint.py:16-94).

An upstream ``*.pt`` is a pickle of ``{"model": <models.yolo.Model nn.Module>, ...}``.  The reference
unpickles it with its vendored copy of the upstream classes (yolort/v5, yolort/v5/helper.py:49-82).
Here no upstream code is vendored: every class the pickle names outside torch / numpy / builtins is
replaced, while unpickling, by an empty ``nn.Module`` stub that just receives the pickled ``__dict__``
(parameters, buffers, sub-modules, ``yaml``, ``stride``).  Nothing from the file is executed beyond
an explicit allow-list of (module, name) data constructors -- the tensor rebuild functions of torch itself,
OrderedDict, numpy's array reconstructors; dotted names are refused.  Layer indices are then renamed to the yolort layout
(index maps: reference _checkpoint.py:53-64) and, like the reference (:81), the weights come back
fp16-rounded.
"""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict

import torch
from torch import nn

_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "range", "object"}
# The ONLY callables a checkpoint may resolve to real objects: an explicit (module, name) allow-list of data
# constructors.  A module-level trust ("everything in torch._utils") is NOT safe -- torch._utils._import_dotted_name,
# collections._sys, ... lead straight to os.system -- and protocol-4 dotted names ("traceback.linecache.os.getcwd")
# traverse attributes, so any name containing a dot is refused as well.
_SAFE_CALLABLES = {
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.nn.parameter", "Parameter"), ("torch.nn.parameter", "Buffer"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"), ("torch.storage", "_LegacyStorage"),
    ("collections", "OrderedDict"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("_codecs", "encode"),
    ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"),
}
_TORCH_STORAGE_NAMES = {f"{t}Storage" for t in ("Float", "Half", "BFloat16", "Double", "Long", "Int", "Short", "Char", "Byte", "Bool")}


class _UpstreamStub(nn.Module):
    """stand-in for any upstream class (models.yolo.Model, models.common.Conv, ...): state only, no code"""

    def __init__(self, *args, **kwargs):   # a pickle may "call" the class with arguments (REDUCE): they are dropped
        super().__init__()

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise RuntimeError("upstream modules are not executable here; only their state is read")


class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        """Resolves ONLY: safe builtins, the allow-listed data constructors above, torch dtypes / legacy storage classes and
        the parameter containers of torch.nn.modules (nn.Module subclasses, which the upstream tree is made of).  Everything
        else the pickle names -- upstream classes, but also any callable an untrusted file smuggles in, however it is
        spelled -- becomes an inert nn.Module stub; nothing from the file is ever imported or executed."""
        if "." not in name:
            if module in ("builtins", "__builtin__"):
                if name in _SAFE_BUILTINS:
                    return super().find_class(module, name)
            elif (module, name) in _SAFE_CALLABLES:
                return super().find_class(module, name)
            elif module == "torch" and (name in _TORCH_STORAGE_NAMES or isinstance(getattr(torch, name, None), torch.dtype)):
                return getattr(torch, name)
            elif module.startswith("torch.nn.modules."):
                obj = super().find_class(module, name)
                if isinstance(obj, type) and issubclass(obj, nn.Module):
                    return obj
        return type(name.replace(".", "_"), (_UpstreamStub,), {"__module__": module})


class _StubPickle:
    """minimal `pickle_module` for torch.load"""

    __name__ = "yolort_amd_stub_pickle"
    Unpickler = _StubUnpickler
    load = staticmethod(lambda f, **kw: _StubUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _StubUnpickler(io.BytesIO(b), **kw).load())
    # torch.load probes these
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PicklingError = pickle.PicklingError
    UnpicklingError = pickle.UnpicklingError


def get_yolov5_size(depth_multiple: float, width_multiple: float) -> str:
    table = {(0.33, 0.25): "n", (0.33, 0.5): "s", (0.67, 0.75): "m", (1.0, 1.0): "l", (1.33, 1.25): "x"}
    key = (round(float(depth_multiple), 2), round(float(width_multiple), 2))
    if key not in table:
        raise NotImplementedError(f"unknown yolov5 size: depth_multiple={depth_multiple}, width_multiple={width_multiple}")
    return table[key]


def _index_maps(use_p6: bool):
    """upstream layer index -> yolort module path (reference _checkpoint.py:53-64)"""
    m = {str(i): f"backbone.body.{i}" for i in range(9)}
    if use_p6:
        m.update({"9": "backbone.pan.intermediate_blocks.p6.0", "10": "backbone.pan.intermediate_blocks.p6.1"})
        inner = {"0": "11", "1": "12", "3": "15", "4": "16", "6": "19", "7": "20"}
        layer = {"0": "23", "1": "24", "2": "26", "3": "27", "4": "29", "5": "30", "6": "32"}
        head = "33"
    else:
        inner = {"0": "9", "1": "10", "3": "13", "4": "14"}
        layer = {"0": "17", "1": "18", "2": "20", "3": "21", "4": "23"}
        head = "24"
    m.update({v: f"backbone.pan.inner_blocks.{k}" for k, v in inner.items()})
    m.update({v: f"backbone.pan.layer_blocks.{k}" for k, v in layer.items()})
    return m, head


def load_from_ultralytics(checkpoint_path: str, version: str = "r6.0") -> Dict[str, Any]:
    if version not in ["r3.1", "r4.0", "r6.0"]:
        raise NotImplementedError(f"Currently does not support version: {version}.")
    # (the layer-index maps below are the same for every release: upstream's r3.1 / r4.0 / r6.0 P5 yamls all place the neck's first block at index 9 and Detect at 24)
    ckpt = torch.load(checkpoint_path, map_location="cpu", pickle_module=_StubPickle, weights_only=False)
    model = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    if isinstance(ckpt, dict) and ckpt.get("ema") is not None and not isinstance(ckpt.get("ema"), (int, float)):
        model = ckpt["ema"]   # upstream prefers the EMA weights when present
    yaml = model.yaml
    layers = model.model      # nn.Sequential stub of the upstream layers, children named "0".."N"
    detect = list(layers.children())[-1]
    strides = [int(s) for s in torch.as_tensor(model.stride).tolist()]
    use_p6 = len(strides) == 4
    num_anchors = detect.anchors.shape[1]
    anchor_grids = (detect.anchors.float() * torch.as_tensor(detect.stride).float().view(-1, 1, 1)).reshape(1, -1, 2 * num_anchors).tolist()[0]
    index_map, head_idx = _index_maps(use_p6)

    src = model.float().state_dict()
    out: Dict[str, torch.Tensor] = {}
    for key, value in src.items():
        parts = key.split(".")
        if parts[0] != "model":
            continue
        idx, rest = parts[1], parts[2:]
        if idx == head_idx:
            if rest[0] == "m":  # Detect.m.{i}.{weight,bias}
                out["head.head." + ".".join(rest[1:])] = value
            continue            # anchors / anchor_grid buffers are rebuilt from strides + anchor_grids
        if idx not in index_map:
            continue            # Upsample / Concat carry no parameters
        out[index_map[idx] + "." + ".".join(rest)] = value
    state_dict = {k: v.half() if v.is_floating_point() else v for k, v in out.items()}
    depth_multiple, width_multiple = yaml["depth_multiple"], yaml["width_multiple"]
    return {
        "num_classes": int(yaml["nc"]),
        "depth_multiple": depth_multiple,
        "width_multiple": width_multiple,
        "strides": strides,
        "anchor_grids": anchor_grids,
        "use_p6": use_p6,
        "size": get_yolov5_size(depth_multiple, width_multiple),
        "state_dict": state_dict,
    }
