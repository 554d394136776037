"""YOLOv5 building blocks (r3.1 / r4.0 / r6.0 forms) as HIP plan emitters.

Same constructor signatures, attribute names and state_dict keys as the reference's blocks
(yolort/v5/models/common.py: Conv :42, Bottleneck :94, BottleneckCSP :119, C3 :149, SPP :176, SPPF :190, Focus :210) so that its
checkpoints load unchanged; the arithmetic is the fused implicit-GEMM kernel (csrc/conv_igemm.hip)
instead of conv2d -> BatchNorm2d -> SiLU -> cat.
"""
from __future__ import annotations

import os

from typing import Dict, Optional, Tuple

import torch
from torch import nn

from ..._lib import ACT_HARDSWISH, ACT_LEAKY, ACT_NONE, ACT_SILU, YmiError
from ...engine import PackedConv, Plan, View
from ...hipmodule import HipModule

__all__ = ["Conv", "Bottleneck", "BottleneckCSP", "C3", "SPP", "SPPF", "Focus", "autopad", "focus_transform", "space_to_depth"]

BN_EPS, BN_MOMENTUM = 1e-3, 0.03  # set after construction by the reference (darknetv6.py:110-112)


def autopad(k, p=None):
    """'same' padding (reference common.py:35-39)."""
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def act_code(act: nn.Module) -> int:
    """the epilogue's activation code (include/yolort_amd.h YMI_ACT_*) of the activation modules the reference's blocks use"""
    if isinstance(act, nn.SiLU):
        return ACT_SILU
    if isinstance(act, nn.Identity):
        return ACT_NONE
    if isinstance(act, nn.Hardswish):
        return ACT_HARDSWISH
    if isinstance(act, nn.LeakyReLU) and abs(act.negative_slope - 0.1) < 1e-12:
        return ACT_LEAKY
    raise NotImplementedError(f"activation {act!r} is not fused (SiLU, Hardswish, LeakyReLU(0.1) and identity are)")


class Conv(HipModule):
    """conv2d (no bias) + BatchNorm2d + SiLU (r4.0 / r6.0) or Hardswish (r3.1), fused (reference common.py:42-73)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True, version="r4.0"):
        super().__init__()
        if g != 1:
            raise NotImplementedError("grouped convolutions are not on the YOLOv5 hot path")
        if version not in ("r4.0", "r3.1"):
            raise NotImplementedError(f"Currently doesn't support version {version}.")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=BN_EPS, momentum=BN_MOMENTUM)
        default = nn.SiLU() if version == "r4.0" else nn.Hardswish()   # reference :62-65
        self.act = default if act is True else (act if isinstance(act, nn.Module) else nn.Identity())
        act_code(self.act)   # refuses activations that no epilogue carries
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
        return plan.conv(x, pc, self.conv.stride, self.conv.padding, act_code(self.act), out=out, res=res, name=name, up2_out=up2_out, chain=chain)


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

    @property
    def out_channels(self) -> int:
        return self.cv3.conv.out_channels

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
        # round 6 (csrc/c3_tile.hip): hidden widths 64 / 128 on maps whose halo strip of whole rows fits the LDS patch -- the whole block in one launch, or
        # one launch per Bottleneck (HEAD, MID ..., TAIL); intermediates of a launch never reach memory, weights stream in fragment order
        if (nb >= 1 and self.cv3.conv.out_channels == 2 * c_ and getattr(plan, "c3_tile_on", False) and plan.c3_tile_ok(x, c_)
                and all(isinstance(b, Bottleneck) and b.cv1.conv.kernel_size == (1, 1) and b.cv2.conv.kernel_size == (3, 3) and b.cv2.conv.stride == (1, 1)
                        and b.cv2.conv.groups == 1 and b.cv1.conv.out_channels == c_ and b.cv2.conv.out_channels == c_ and b.add == self.m[0].add
                        and isinstance(b.cv1.act, nn.SiLU) and isinstance(b.cv2.act, nn.SiLU) for b in self.m)
                and all(isinstance(c.act, nn.SiLU) for c in (self.cv1, self.cv2, self.cv3)) and (out is None or out.cs % 8 == 0)):
            pk = lambda cv, cin: cv.packed(plan.dtype, plan.device, cin)  # noqa: E731
            pc12, pc3 = self.packed_pair(plan.dtype, plan.device, x.c), pk(self.cv3, 2 * c_)
            add = bool(self.m[0].add)
            if nb == 1:
                return plan.c3_tile(0, pk(self.m[0].cv1, c_), pk(self.m[0].cv2, c_), add, x=x, pc12=pc12, pc3=pc3, out=out, name=name + ".tile")
            y2 = plan.alloc(x.n, x.h, x.w, c_)
            ping = [plan.alloc(x.n, x.h, x.w, c_), plan.alloc(x.n, x.h, x.w, c_) if nb > 2 else None]
            cur = plan.c3_tile(1, pk(self.m[0].cv1, c_), pk(self.m[0].cv2, c_), add, x=x, pc12=pc12, y1_out=ping[0], y2=y2, name=name + ".tile.cv1+cv2+m.0")
            for j in range(1, nb - 1):
                cur = plan.c3_tile(2, pk(self.m[j].cv1, c_), pk(self.m[j].cv2, c_), add, y1_in=cur, y1_out=ping[j % 2], name=f"{name}.tile.m.{j}")
            return plan.c3_tile(3, pk(self.m[nb - 1].cv1, c_), pk(self.m[nb - 1].cv2, c_), add, y1_in=cur, y2=y2, pc3=pc3, out=out, name=f"{name}.tile.m.{nb - 1}+cv3")
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


class BottleneckCSP(HipModule):
    """cv4(LeakyReLU(BN(cat(cv3(m(cv1(x))), cv2(x))))) (reference :119-146; the r3.1 block: Hardswish in its Conv modules, LeakyReLU(0.1) after the shared BatchNorm).
    `cv2` and `cv3` are bare convolutions whose outputs meet in one BatchNorm over the concatenation: its first half is folded into cv3, its second into cv2, each
    followed by the LeakyReLU launch (Plan.conv with a legacy activation = convolution + ymi_act), both writing straight into the concat buffer cv4 reads."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1, version="r3.1")
        self.cv2 = nn.Conv2d(c1, c_, 1, 1, bias=False)
        self.cv3 = nn.Conv2d(c_, c_, 1, 1, bias=False)
        self.cv4 = Conv(2 * c_, c2, 1, 1, version="r3.1")
        self.bn = nn.BatchNorm2d(2 * c_, eps=BN_EPS, momentum=BN_MOMENTUM)
        self.act = nn.LeakyReLU(0.1, inplace=True)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0, version="r3.1") for _ in range(n)])
        self._half: Dict[Tuple, Tuple] = {}

    @property
    def out_channels(self) -> int:
        return self.cv4.conv.out_channels

    def packed_half(self, which: int, dtype: torch.dtype, device: torch.device, cin_view: int) -> PackedConv:
        """cv3 (which = 0) / cv2 (which = 1) with its half of the shared BatchNorm folded in"""
        conv = self.cv3 if which == 0 else self.cv2
        c_ = conv.out_channels
        sl = slice(which * c_, (which + 1) * c_)
        bnp = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        sig = tuple(t._version for t in (conv.weight,) + bnp) + (conv.weight.data_ptr(),)
        key = (which, dtype, device, cin_view)
        hit = self._half.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        pc = PackedConv(conv.weight, None, tuple(t[sl] for t in bnp) + (float(self.bn.eps),), dtype, device, cin_pad=cin_view)
        self._half[key] = (sig, pc)
        return pc

    def emit(self, plan: Plan, x: View, out: Optional[View] = None, name: str = "csp") -> View:
        c_ = self.cv1.conv.out_channels
        if c_ % 8:
            raise YmiError("BottleneckCSP hidden width must be a multiple of 8")
        cat = plan.alloc(x.n, x.h, x.w, 2 * c_)
        y = self.cv1.emit(plan, x, name=name + ".cv1")
        for j, b in enumerate(self.m):
            y = b.emit(plan, y, name=f"{name}.m.{j}")
        act = act_code(self.act)
        plan.conv(y, self.packed_half(0, plan.dtype, plan.device, y.c), 1, 0, act, out=cat.slice_c(0, c_), name=name + ".cv3+bn")
        plan.conv(x, self.packed_half(1, plan.dtype, plan.device, x.c), 1, 0, act, out=cat.slice_c(c_, c_), name=name + ".cv2+bn")
        return self.cv4.emit(plan, cat, out=out, name=name + ".cv4")


def focus_transform(x):
    """x(b,c,h,w) -> y(b,4c,h/2,w/2) (reference :237-240; the torch expression, for callers of the reference's helper -- the HIP path never materialises it)"""
    return torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)


def space_to_depth(x):
    """reference :243-249: the same rearrangement through view / permute"""
    n, c, h, w = x.size()
    x = x.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 5, 3, 1, 2, 4)
    return x.reshape(n, 4 * c, h // 2, w // 2)


class Focus(HipModule):
    """Conv(4 c1, c2, k) over the space-to-depth rearrangement of the image (reference :210-234), the stem of the r3.1 / r4.0 models.
    The rearrangement is never materialised: slot (dy, dx) of `focus_transform` holds pixel (2Y + dy, 2X + dx), so a k x k convolution with 'same' padding over the 12
    half-resolution channels IS a 2k x 2k stride-2 convolution over the image -- for the reference's k = 3: Conv(3, c2, 6, 2, 2) with
    W6[o, c, 2 ky + dy, 2 kx + dx] = W3[o, 3 slot(dy, dx) + c, ky, kx], slot = (0,0) (1,0) (0,1) (1,1) -> 0 1 2 3 (ultralytics/yolov5#4825, the equivalence the r6.0 stem was
    introduced with).  Same products, same zero padding, another summation order: the stem kernels (super-pixel form, planar-image form, fused stem + body.1) run unchanged."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True, version="r4.0"):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act, version=version)
        self._packed: Dict[Tuple, Tuple] = {}

    def _input_cpad(self, c: int) -> int:
        return 4

    def stem_weight(self) -> torch.Tensor:
        w = self.conv.conv.weight   # (c2, 4 c1, 3, 3)
        c2, c4, kh, kw = w.shape
        c1 = c4 // 4
        if (kh, kw) != (3, 3) or self.conv.conv.stride != (1, 1) or self.conv.conv.padding != (1, 1) or c1 != 3:
            raise NotImplementedError("Focus is emitted for the reference's stem only: 3 image channels, k = 3, s = 1, 'same' padding")
        w6 = w.new_zeros(c2, c1, 6, 6)
        for slot, (dy, dx) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
            w6[:, :, dy::2, dx::2] = w[:, slot * c1:(slot + 1) * c1]
        return w6

    def packed(self, dtype: torch.dtype, device: torch.device) -> PackedConv:
        cv = self.conv
        sig = tuple(t._version for t in (cv.conv.weight, cv.bn.weight, cv.bn.bias, cv.bn.running_mean, cv.bn.running_var)) + (cv.conv.weight.data_ptr(),)
        key = (dtype, device)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        bn = (cv.bn.weight, cv.bn.bias, cv.bn.running_mean, cv.bn.running_var, float(cv.bn.eps))
        pc = PackedConv(self.stem_weight(), None, bn, dtype, device, cin_pad=None, stem_superpixel=True)
        self._packed[key] = (sig, pc)
        return pc

    def emit(self, plan: Plan, x: View, out: Optional[View] = None, name: str = "focus") -> View:
        if x.c != 4:
            raise YmiError("Focus reads the NHWC4 image view (it is the network's first layer)")
        return plan.conv(x, self.packed(plan.dtype, plan.device), (2, 2), (2, 2), act_code(self.conv.act), out=out, name=name)


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
