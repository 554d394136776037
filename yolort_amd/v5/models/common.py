"""YOLOv5 building blocks (r4.0/r6.0 forms) as HIP plan emitters.

Same constructor signatures, attribute names and state_dict keys as the reference's blocks
(yolort/v5/models/common.py: Conv :42, Bottleneck :94, C3 :149, SPP :176, SPPF :190) so that its
checkpoints load unchanged; the arithmetic is the fused implicit-GEMM kernel (csrc/conv_igemm.hip)
instead of conv2d -> BatchNorm2d -> SiLU -> cat.
"""
from __future__ import annotations

import os

from typing import Dict, Optional, Tuple

import torch
from torch import nn

from ..._lib import ACT_NONE, ACT_SILU, YmiError
from ...engine import PackedConv, Plan, View
from ...hipmodule import HipModule

__all__ = ["Conv", "Bottleneck", "C3", "SPP", "SPPF", "autopad"]

BN_EPS, BN_MOMENTUM = 1e-3, 0.03  # set after construction by the reference (darknetv6.py:110-112)


def autopad(k, p=None):
    """'same' padding (reference common.py:35-39)."""
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class Conv(HipModule):
    """conv2d (no bias) + BatchNorm2d + SiLU, fused (reference common.py:42-73)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True, version="r4.0"):
        super().__init__()
        if g != 1:
            raise NotImplementedError("grouped convolutions are not on the YOLOv5 r6.0 hot path")
        if version != "r4.0":
            raise NotImplementedError(f"Currently doesn't support version {version}.")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=BN_EPS, momentum=BN_MOMENTUM)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())
        if not isinstance(self.act, (nn.SiLU, nn.Identity)):
            raise NotImplementedError("only SiLU / identity activations are fused")
        self._packed: Dict[Tuple, Tuple] = {}

    def _is_stem(self) -> bool:
        return self.conv.in_channels == 3 and self.conv.kernel_size == (6, 6) and self.conv.stride == (2, 2) and self.conv.padding == (2, 2)

    def _input_cpad(self, c: int) -> int:
        return 4 if self._is_stem() else (c + 7) // 8 * 8

    def packed(self, dtype: torch.dtype, device: torch.device, cin_view: int) -> PackedConv:
        sig = tuple(t._version for t in (self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)) + (self.conv.weight.data_ptr(),)
        stem = cin_view == 4 and self._is_stem()
        key = (dtype, device, cin_view, stem)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        bn = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, float(self.bn.eps))
        pc = PackedConv(self.conv.weight, None, bn, dtype, device, cin_pad=None if stem else cin_view, stem_superpixel=stem)
        self._packed[key] = (sig, pc)
        return pc

    def emit(self, plan: Plan, x: View, out: Optional[View] = None, res: Optional[View] = None, name: str = "conv", up2_out: Optional[View] = None,
             chain=None) -> View:
        pc = self.packed(plan.dtype, plan.device, x.c)
        act = ACT_SILU if isinstance(self.act, nn.SiLU) else ACT_NONE
        return plan.conv(x, pc, self.conv.stride, self.conv.padding, act, out=out, res=res, name=name, up2_out=up2_out, chain=chain)


class Bottleneck(HipModule):
    """x + cv2(cv1(x)) (1x1 then 3x3); the residual add rides in cv2's epilogue (reference :94-116)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5, version="r4.0"):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1, version=version)
        self.cv2 = Conv(c_, c2, 3, 1, g=g, version=version)
        self.add = shortcut and c1 == c2

    def emit(self, plan: Plan, x: View, out: Optional[View] = None, name: str = "bottleneck", t: Optional[View] = None, chain=None, chain_name: Optional[str] = None) -> View:
        """`t`: cv1's output when the producer of x already computed it (chained 1x1, see C3.emit);
        `chain`: a conv chained to cv2's output (C3.cv3 over the concat), passed through to Plan.conv"""
        if t is not None:
            y = t
        else:
            c_ = self.cv1.conv.out_channels
            # a hidden width that is not a multiple of 32 (yolov5m: 48 at 320 x 320) would send the 3x3 through the im2col-TABLE form of the
            # implicit GEMM (k32 operand chunks straddle taps: 0.13 of its bound, profiles/r02z_layer_table_c3.csv).  Instead cv1 writes its
            # c_ channels into a zero-initialised buffer of round_up(c_, 32) channels and the 3x3 reads all of them against weights whose
            # extra K rows are zero: the unit-tap kernels (LDS halo, resident weights) apply, for a third more bytes on this one tensor.
            cp = (c_ + 31) // 32 * 32
            if cp != c_ and c_ > 32 and not plan.fp32 and os.environ.get("YOLORT_AMD_PAD_HIDDEN", "1") != "0" and self.cv2.conv.groups == 1:
                tb = plan.alloc(x.n, x.h, x.w, cp, zero=True)
                self.cv1.emit(plan, x, out=tb.slice_c(0, c_), name=name + ".cv1")
                y = tb
            else:
                y = self.cv1.emit(plan, x, name=name + ".cv1")
        tag = ".cv2"
        if chain is not None:
            tag = ".cv2+" + (chain_name or "cv3")
        return self.cv2.emit(plan, y, out=out, res=x if self.add else None, name=name + tag, chain=chain)


class C3(HipModule):
    """cv3(cat(m(cv1(x)), cv2(x))) (reference :149-173).  The concat buffer is allocated first; the
    last bottleneck and cv2 write straight into its two halves, so no cat kernel exists."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, version="r4.0"):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1, version=version)
        self.cv2 = Conv(c1, c_, 1, 1, version=version)
        self.cv3 = Conv(2 * c_, c2, 1, version=version)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0, version=version) for _ in range(n)])
        self._pair: Dict[Tuple, Tuple] = {}

    def packed_pair(self, dtype: torch.dtype, device: torch.device, cin_view: int) -> PackedConv:
        """cv1 and cv2 read the same input: their folded weights are stacked along cout so ONE launch
        computes both (the kernel's second-output feature routes cv2's half into the concat buffer)."""
        mods = (self.cv1, self.cv2)
        sig = tuple(t._version for m in mods for t in (m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var)) + (self.cv1.conv.weight.data_ptr(),)
        key = (dtype, device, cin_view)
        hit = self._pair.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        cat = lambda f: torch.cat([f(m) for m in mods])  # noqa: E731
        bn = (cat(lambda m: m.bn.weight), cat(lambda m: m.bn.bias), cat(lambda m: m.bn.running_mean), cat(lambda m: m.bn.running_var), float(self.cv1.bn.eps))
        pc = PackedConv(cat(lambda m: m.conv.weight), None, bn, dtype, device, cin_pad=cin_view)
        self._pair[key] = (sig, pc)
        return pc

    def emit(self, plan: Plan, x: View, out: Optional[View] = None, name: str = "c3") -> View:
        c_ = self.cv1.conv.out_channels
        nb = len(self.m)
        # default (YOLORT_AMD_FUSE_C3=0 turns it off, engine.Plan.fuse_c3): the whole block in one launch when it is the instance
        # csrc/c3_fused32.hip holds -- 64 -> 64, one shortcut Bottleneck of 32 hidden channels (yolov5s backbone.body.2)
        if (getattr(plan, "fuse_c3", False) and not plan.use_v1 and nb == 1 and c_ == 32 and x.c == 64 and self.cv3.conv.out_channels == 64
                and isinstance(self.m[0], Bottleneck) and self.m[0].add and self.m[0].cv1.conv.kernel_size == (1, 1) and self.m[0].cv2.conv.kernel_size == (3, 3)
                and self.m[0].cv2.conv.groups == 1
                and all(isinstance(c.act, nn.SiLU) for c in (self.cv1, self.cv2, self.cv3, self.m[0].cv1, self.m[0].cv2))
                and (out is None or out.cs % 8 == 0) and x.cs % 8 == 0):
            b0 = self.m[0]
            return plan.c3_fused(x, self.packed_pair(plan.dtype, plan.device, x.c), b0.cv1.packed(plan.dtype, plan.device, c_),
                                 b0.cv2.packed(plan.dtype, plan.device, c_), self.cv3.packed(plan.dtype, plan.device, 2 * c_), out=out, name=name + ".fused")
        cat = plan.alloc(x.n, x.h, x.w, 2 * c_)
        fuse = (not plan.use_v1) and c_ % 8 == 0 and nb >= 1 and isinstance(self.cv1.act, nn.SiLU) and isinstance(self.cv2.act, nn.SiLU)
        t0 = None
        if fuse:
            y = plan.alloc(x.n, x.h, x.w, c_)
            b0 = self.m[0]
            # the first Bottleneck's 1x1 (cv1) rides in the same launch: its input is this conv's freshly rounded output
            chain_ok = (plan.chain_1x1 and c_ in (32, 64) and   # (K = 128 works in the kernels but costs the main conv more than the saved launch: measured in round 4, not emitted)
                         isinstance(b0, Bottleneck) and isinstance(b0.cv1.act, nn.SiLU) and b0.cv1.conv.kernel_size == (1, 1)
                        and b0.cv1.conv.out_channels % 32 == 0 and b0.cv1.conv.out_channels <= 128)
            chain = None
            if chain_ok:
                t0 = plan.alloc(x.n, x.h, x.w, b0.cv1.conv.out_channels)
                chain = (b0.cv1.packed(plan.dtype, plan.device, c_), t0)
            plan.conv(x, self.packed_pair(plan.dtype, plan.device, x.c), 1, 0, ACT_SILU, out=y, out2=cat.slice_c(c_, c_), split=c_,
                      name=name + (".cv1+cv2+m.0.cv1" if chain_ok else ".cv1+cv2"), chain=chain)
        else:
            y = self.cv1.emit(plan, x, out=cat.slice_c(0, c_) if nb == 0 else None, name=name + ".cv1")
        # cv3 over the concat [last Bottleneck output | cv2(x)] rides in the last Bottleneck's 3x3 launch: its first K range is
        # that launch's rounded output in registers, the second is read from the concat buffer's other half
        c2 = self.cv3.conv.out_channels
        chain3 = None
        if (fuse and plan.chain_1x1 and plan.chain_cv3 and nb >= 1 and c_ in (32, 64) and c2 % 32 == 0 and c2 <= 128 and isinstance(self.cv3.act, nn.SiLU)
                and isinstance(self.m[nb - 1], Bottleneck) and self.m[nb - 1].cv2.conv.kernel_size == (3, 3) and self.m[nb - 1].cv2.conv.out_channels == c_):
            o3 = out if out is not None else plan.alloc(x.n, x.h, x.w, c2)
            chain3 = (self.cv3.packed(plan.dtype, plan.device, 2 * c_), o3, cat.slice_c(c_, c_))
        t_next = None
        for j, b in enumerate(self.m):
            kw = {}
            if j == 0 and t0 is not None:
                kw["t"] = t0
            elif t_next is not None:
                kw["t"] = t_next
            t_next = None
            if j == nb - 1 and chain3 is not None:
                kw["chain"] = chain3
            elif (j < nb - 1 and getattr(plan, "chain_next", False) and (plan.chain_next is True or plan.chain_next == c_) and not plan.use_v1 and c_ in (32, 64, 128) and isinstance(b, Bottleneck) and isinstance(self.m[j + 1], Bottleneck)
                  and b.cv2.conv.kernel_size == (3, 3) and b.cv2.conv.stride == (1, 1) and b.cv2.conv.groups == 1 and b.cv2.conv.out_channels == c_
                  and isinstance(b.cv2.act, nn.SiLU) and isinstance(self.m[j + 1].cv1.act, nn.SiLU) and self.m[j + 1].cv1.conv.kernel_size == (1, 1)
                  and self.m[j + 1].cv1.conv.out_channels % 32 == 0 and self.m[j + 1].cv1.conv.out_channels <= 128):
                # this Bottleneck's 3x3 carries the next one's 1x1 in its epilogue (reference :115-116 twice: x + cv2(cv1(x)), then cv1 of the next)
                nxt = self.m[j + 1]
                t_next = plan.alloc(x.n, x.h, x.w, nxt.cv1.conv.out_channels)
                kw["chain"] = (nxt.cv1.packed(plan.dtype, plan.device, c_), t_next)
                kw["chain_name"] = f"m.{j + 1}.cv1"
            y = b.emit(plan, y, out=cat.slice_c(0, c_) if j == nb - 1 else None, name=f"{name}.m.{j}", **kw)
        if not fuse:
            self.cv2.emit(plan, x, out=cat.slice_c(c_, c_), name=name + ".cv2")
        if chain3 is not None:
            return chain3[1]
        return self.cv3.emit(plan, cat, out=out, name=name + ".cv3")


class SPP(HipModule):
    """cv2(cat(x, mp5(x), mp9(x), mp13(x))) after cv1 (reference :176-187)."""

    def __init__(self, c1, c2, k=(5, 9, 13), version="r4.0"):
        super().__init__()
        if tuple(k) != (5, 9, 13):
            raise NotImplementedError("the fused pool pyramid implements k=(5, 9, 13)")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1, version=version)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1, version=version)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def emit(self, plan: Plan, x: View, out: Optional[View] = None, name: str = "spp") -> View:
        c_ = self.cv1.conv.out_channels
        if c_ % 8:
            raise YmiError("SPP hidden width must be a multiple of 8")
        cat = plan.alloc(x.n, x.h, x.w, 4 * c_)
        self.cv1.emit(plan, x, out=cat.slice_c(0, c_), name=name + ".cv1")
        plan.spp_pool(cat, c_, name=name + ".pool")
        return self.cv2.emit(plan, cat, out=out, name=name + ".cv2")


class SPPF(SPP):
    """SPPF(k=5) == SPP(k=(5,9,13)) (reference :190-207, equivalence note :196); same parameters."""

    def __init__(self, c1, c2, k=5, version="r4.0"):
        if k != 5:
            raise NotImplementedError("the fused pool pyramid implements SPPF(k=5)")
        super().__init__(c1, c2, (5, 9, 13), version=version)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
