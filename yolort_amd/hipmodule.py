"""Base class of every graph piece: parameters live in ordinary torch containers (so state_dict
keys match the reference's checkpoints), compute is emitted into a HIP plan (engine.Plan).

`forward` of any such module accepts the reference's NCHW tensors, converts at the edge, runs the
module's own plan on the MI355X and converts back -- that is how the reference's per-block shape
tests (test/test_v5_common.py, test/test_models.py:188-274) run against this package.  There is no
eager/CPU implementation: off-GPU `forward` raises.
"""
from __future__ import annotations

import functools
import itertools
import operator
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor, nn

from . import _lib
from ._lib import YmiError, check, dtype_code
from .engine import Plan, View


def compute_dtype_of(module: nn.Module) -> torch.dtype:
    """fp16/bf16 models compute in their own dtype; fp32-parameter models compute in `module.compute_dtype`: fp16 by
    default (16-bit storage, fp32 accumulation), or torch.float32 = the PARITY MODE of csrc/conv_f32.hip (fp32 storage and
    exact fp32 MFMA arithmetic; ~16x slower, reproduces the fp32 CPU reference to rounding-order accuracy)."""
    p = next(module.parameters(), None)
    if p is not None and p.dtype in (torch.float16, torch.bfloat16):
        return p.dtype
    return getattr(module, "compute_dtype", torch.float16)


_VERSION_OF = operator.attrgetter("_version")
_DATA_PTR_OF = operator.methodcaller("data_ptr")


def _values_of(dicts):
    return itertools.chain.from_iterable(map(dict.values, dicts))


def _load_sig_ext():
    """yolort_amd/lib/_ymi_sig.so (torch_ext/sig_ext.cpp, built by `python -m yolort_amd.torch_ext` / __graft_entry__.build()): the walk below as one C call.
    Optional; YOLORT_AMD_SIG_EXT=0 keeps the interpreter-level walk."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_ymi_sig.so")
    if os.environ.get("YOLORT_AMD_SIG_EXT", "1") == "0" or not os.path.exists(path):
        return None
    try:
        spec = importlib.util.spec_from_file_location("_ymi_sig", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except Exception:   # built against another torch / python: the Python walk is always there
        return None


_SIG_EXT = _load_sig_ext()


def _collect(module: nn.Module):
    """the dict OBJECTS of the tree, split by whether they hold anything: the values of the non-empty ones are re-read on every call, of the empty ones only the lengths"""
    mods = list(module.modules())
    t_all = [d for m in mods for d in (m._parameters, m._buffers)]
    m_all = [m._modules for m in mods]
    t_full, t_empty = [d for d in t_all if d], [d for d in t_all if not d]
    m_full, m_empty = [d for d in m_all if d], [d for d in m_all if not d]
    kids = _SIG_EXT.scan(t_full, m_full, t_empty, m_empty)[3] if _SIG_EXT is not None else tuple(map(id, _values_of(m_full)))
    # (`mods`: the module objects themselves stay referenced for as long as this collection is the cache -- `kids` compares ADDRESSES, and after `del m.sub; m.sub = New()`
    # CPython likes to hand the new instance of the same class the address that was just freed: with the old child still alive here that cannot happen.  ADVICE r5)
    return t_full, t_empty, m_full, m_empty, kids, mods


def weights_signature(module: nn.Module) -> Tuple:
    """The plan cache's key: changes when any parameter / buffer of the module tree is modified in place (`_version`), re-allocated (`data_ptr`: .to() / .half() swap the
    data of the same Parameter object), replaced, removed or set to None, or when a sub-module is added, replaced or deleted.

    Called once per submitted batch.  Every call re-reads the LIVE `_parameters` / `_buffers` / `_modules` dicts of the module tree (the dict OBJECTS are cached, their
    values are not): the identities of all registered tensors and child modules, then `_version` and `data_ptr()` of every tensor.  Nothing is inferred from registration
    hooks -- round 4 cached the tensor OBJECTS and refreshed them only when one of three process-wide nn.Module registration hooks fired, which `del model.sub[0]`
    (`__delattr__` fires no hook), `m.bias = None` (`register_parameter(None)` fires none) and `_apply` under `torch.__future__.set_overwrite_module_params_on_conversion(True)`
    (new Parameters written straight into `_parameters`) all went past; the hooks are gone.  With `_ymi_sig.so` built (torch_ext/sig_ext.cpp) the walk is ONE C
    call over the cached dict objects (PyDict_Next + the tensors' version counters and data pointers read in C++): ~0.01 ms on yolov5s (275 modules, 348 tensors).  Without
    it the same walk runs as interpreter-level iterator chains (map / chain / reduce), 0.14 ms.  Either way dicts that were empty at collection time are only asked for their
    length.  `YOLO.freeze_weights()` is the only mode that skips this validation (the caller promises not to touch the weights).  (An in-place update through `param.data`
    does not move `_version` -- torch's own rule.)"""
    cache = module.__dict__.get("_ymi_sig_cache")
    if _SIG_EXT is not None:
        if cache is not None:
            ids, versions, ptrs, kids, stray = _SIG_EXT.scan(cache[0], cache[2], cache[1], cache[3])
            if kids != cache[4] or stray:
                cache = None   # a child was added / replaced / deleted, or a module that held nothing got a tensor or a child: the set of dicts itself is stale
        if cache is None:
            cache = module.__dict__["_ymi_sig_cache"] = _collect(module)
            ids, versions, ptrs, kids, stray = _SIG_EXT.scan(cache[0], cache[2], cache[1], cache[3])
        held = module.__dict__.get("_ymi_sig_tensors")
        if held is None or held[0] != ids:   # keep the very objects of this call alive until the next one: an address in `ids` cannot be recycled by a different object meanwhile
            module.__dict__["_ymi_sig_tensors"] = (ids, list(_values_of(cache[0])))
        return (ids, versions, ptrs)
    if cache is not None:
        t_full, t_empty, m_full, m_empty, child_ids = cache[:5]
        if tuple(map(id, _values_of(m_full))) != child_ids or sum(map(len, m_empty)) or sum(map(len, t_empty)):
            cache = None
    if cache is None:
        cache = module.__dict__["_ymi_sig_cache"] = _collect(module)
    t_full = cache[0]
    ids = tuple(map(id, _values_of(t_full)))                 # identities of everything registered right now (None slots included)
    held = module.__dict__.get("_ymi_sig_tensors")
    if held is None or held[0] != ids:                       # the very objects of the last call (the list keeps them alive, so an id cannot have been recycled): reuse the filtered list
        held = module.__dict__["_ymi_sig_tensors"] = (ids, [t for t in _values_of(t_full) if t is not None])
    tensors = held[1]
    return (hash(ids), sum(map(_VERSION_OF, tensors)), functools.reduce(operator.xor, map(_DATA_PTR_OF, tensors), 0))


def nchw_to_view(plan_or_none: Optional[Plan], x: Tensor, c_pad: int, out: Optional[View] = None, dtype: Optional[torch.dtype] = None) -> View:
    lib = _lib.load(require_gpu=True)
    n, c, h, w = x.shape
    x = x.contiguous()
    if out is None:
        t = torch.zeros(n * h * w * c_pad, device=x.device, dtype=dtype or x.dtype)
        out = View(t, 0, n, h, w, c_pad, c_pad)
    check(lib.ymi_nchw_to_nhwc(x.data_ptr(), n, c, h, w, dtype_code(x.dtype), out.ptr, out.cs, c_pad, dtype_code(out.dtype), _lib.stream_ptr()), "ymi_nchw_to_nhwc")
    return out


def view_to_nchw(v: View, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    lib = _lib.load(require_gpu=True)
    y = torch.empty(v.n, v.c, v.h, v.w, device=v.base.device, dtype=out_dtype or v.dtype)
    check(lib.ymi_nhwc_to_nchw(v.ptr, v.cs, v.n, v.c, v.h, v.w, dtype_code(v.dtype), y.data_ptr(), dtype_code(y.dtype), _lib.stream_ptr()), "ymi_nhwc_to_nchw")
    return y


class HipModule(nn.Module):
    """nn.Module whose forward is a cached HIP plan of its own `emit`."""

    def __init__(self) -> None:
        super().__init__()
        self._plans: Dict[Tuple, Tuple] = {}

    def _input_cpad(self, c: int) -> int:
        """channel padding of the NHWC input view built at the NCHW API edge"""
        return (c + 7) // 8 * 8

    # subclasses implement: emit(plan, x: View | List[View], out=None) -> View | List[View]
    def emit(self, plan: Plan, x, out=None):  # pragma: no cover
        raise NotImplementedError

    def _inputs_as_list(self, x) -> Tuple[List[Tensor], bool]:
        if isinstance(x, Tensor):
            return [x], False
        if isinstance(x, dict):
            return list(x.values()), True
        return list(x), True

    def forward(self, x):
        if self.training:
            raise NotImplementedError("yolort_amd implements the inference path only; call .eval() (training is out of scope)")
        xs, is_list = self._inputs_as_list(x)
        for t in xs:
            if not t.is_cuda:
                raise YmiError("yolort_amd runs on an MI355X only: move the model and inputs to 'cuda' (there is no CPU fallback)")
            if t.dim() != 4:
                raise ValueError(f"expected NCHW tensors, got shape {tuple(t.shape)}")
        with torch.cuda.device(xs[0].device):   # plan construction and launches run on the inputs' device
            return self._forward_on_device(xs, is_list)

    def _forward_on_device(self, xs, is_list):
        cdt = compute_dtype_of(self)
        key = (tuple(tuple(t.shape) for t in xs), cdt, xs[0].device.index, weights_signature(self))
        entry = self._plans.get(key)
        if entry is None:
            self._plans.clear()
            plan = Plan(xs[0].device, cdt)
            ins = []
            for t in xs:
                c_pad = self._input_cpad(t.shape[1])
                ins.append(plan.alloc(t.shape[0], t.shape[2], t.shape[3], c_pad, zero=True))
            outs = self.emit(plan, ins if is_list else ins[0])
            entry = (plan, ins, outs)
            self._plans[key] = entry
        plan, ins, outs = entry
        for t, v in zip(xs, ins):
            nchw_to_view(plan, t, v.c, out=v)
        plan.run()
        odt = xs[0].dtype if xs[0].dtype.is_floating_point else torch.float32
        if isinstance(outs, View):
            return view_to_nchw(outs, odt)
        return [view_to_nchw(o, odt) for o in outs]
