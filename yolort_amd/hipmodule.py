"""Base class of every graph piece: parameters live in ordinary torch containers (so state_dict
keys match the reference's checkpoints), compute is emitted into a HIP plan (engine.Plan).

`forward` of any such module accepts the reference's NCHW tensors, converts at the edge, runs the
module's own plan on the MI355X and converts back -- that is how the reference's per-block shape
tests (test/test_v5_common.py, test/test_models.py:188-274) run against this package.  There is no
eager/CPU implementation: off-GPU `forward` raises.
"""
from __future__ import annotations

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


# Structure epoch: bumped whenever ANY nn.Module registers a parameter, a buffer or a sub-module (torch's global registration hooks fire for
# `m.weight = nn.Parameter(...)`, `m.head = other_head`, `register_buffer`, `add_module`, ... -- every path through nn.Module.__setattr__).  While it stands still the
# set of Parameter / buffer OBJECTS of a module tree cannot have changed, so `weights_signature` may keep the flat tensor list it collected (the previous
# cache fingerprinted only modules that had children, and read only modules that already had parameters: a buffer registered later on a bare module, or a child
# added to a former leaf, went unseen; the id() sum could also cancel out).
_STRUCT_EPOCH = [0]


def _bump_epoch(*_args):
    _STRUCT_EPOCH[0] += 1
    return None


nn.modules.module.register_module_parameter_registration_hook(_bump_epoch)
nn.modules.module.register_module_buffer_registration_hook(_bump_epoch)
nn.modules.module.register_module_module_registration_hook(_bump_epoch)


def weights_signature(module: nn.Module) -> Tuple:
    """changes when any parameter / buffer of the module tree is modified in place (`_version`), re-allocated (`data_ptr`: .to() / .half() swap the data of the
    same Parameter object) or replaced (a new object is registered: the structure epoch moves and the tensor list is rebuilt): the plan cache's key.  Called once
    per submitted batch, so it must be cheap: the flat list of tensor objects is cached per module tree and only `_version` / `data_ptr()` of each are read
    (`module.parameters()` + `module.buffers()` walk the tree through two recursive generators with name bookkeeping: 0.4 ms per call on yolov5s, most of round 2's
    0.76 ms per batch -- tools/host_profile.py).  The key holds the identity of the tensor objects, not the epoch: a registration somewhere else in the process (another
    model being built) re-collects the list once and yields the same key.  (Direct pokes into a module's `_parameters` dict bypass nn.Module's registration and are not seen.)"""
    cache = module.__dict__.get("_ymi_tensors")
    if cache is None or cache[0] != _STRUCT_EPOCH[0]:
        mods = list(module.modules())
        tensors = [t for m in mods for t in m._parameters.values() if t is not None] + [t for m in mods for t in m._buffers.values() if t is not None]
        cache = (_STRUCT_EPOCH[0], tensors, hash(tuple(id(t) for t in tensors)))
        module.__dict__["_ymi_tensors"] = cache
    sig = 0
    ptr = 0
    for t in cache[1]:
        sig += t._version
        ptr ^= t.data_ptr()
    return (cache[2], sig, ptr)


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
