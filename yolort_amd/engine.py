"""Plan builder / executor: turns a module tree into a recorded sequence of HIP launches.

The reference dispatches one ATen op per Python call (yolort/models/yolo.py:159-175 walks
backbone -> head -> anchor_generator -> post_process module by module).  Here every module only
*emits* its launches once into a C-side plan (`ymi_plan`, include/yolort_amd.h) for a given
(shape, dtype); a forward is then a single `ymi_plan_run`, optionally a hipGraph replay.

Activations are NHWC views into plan-owned torch buffers; a view may be a channel slice of a wider
buffer (`cstride > c`), which is how every `torch.cat` of the reference (common.py:173,187,
path_aggregation_network.py:224,235) disappears: producers write straight into their slot.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import ACT_NONE, ACT_SILU, C3Desc, ConvDesc, PostDesc, YmiError, check, dtype_code


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class View:
    """NHWC activation view inside a flat torch buffer."""

    base: Tensor  # flat storage tensor (kept alive by the plan)
    off: int      # element offset of channel 0 of pixel 0
    n: int
    h: int
    w: int
    c: int
    cs: int       # pixel stride in elements (>= c)
    tail: int = -1  # element index of the buffer's zero tail (-1: foreign buffer without one)

    @property
    def dtype(self) -> torch.dtype:
        return self.base.dtype

    @property
    def ptr(self) -> int:
        return self.base.data_ptr() + self.off * self.base.element_size()

    def slice_c(self, c0: int, c: int) -> "View":
        assert 0 <= c0 and c0 + c <= self.c
        return View(self.base, self.off + c0, self.n, self.h, self.w, c, self.cs, self.tail)

    def as_tensor(self) -> Tensor:
        """(n,h,w,c) strided torch view of the data (debug / tests / API edges)."""
        return torch.as_strided(self.base, (self.n, self.h, self.w, self.c), (self.h * self.w * self.cs, self.w * self.cs, self.cs, 1), self.off)


class PackedConv:
    """Weights of one convolution, BatchNorm folded, packed [cout_pad][k_pad] K-major for the
    implicit-GEMM kernel (k = (ky*kw + kx)*cin + c), plus fp32 bias.

    BN folding follows yolort/v5/utils/torch_utils.py:238-245: w' = w * g/sqrt(var+eps),
    b' = beta - mean * g/sqrt(var+eps)  (eps = 1e-3, darknetv6.py:110-112).
    """

    def __init__(self, weight: Tensor, bias: Optional[Tensor], bn: Optional[Tuple[Tensor, Tensor, Tensor, Tensor, float]],
                 dtype: torch.dtype, device: torch.device, cin_pad: Optional[int] = None, stem_superpixel: bool = False):
        w = weight.detach().to(device=device, dtype=torch.float32)
        cout, cin, kh, kw = w.shape
        b = torch.zeros(cout, device=device, dtype=torch.float32) if bias is None else bias.detach().to(device=device, dtype=torch.float32)
        if bn is not None:
            g, beta, mean, var, eps = bn
            scale = g.detach().to(device=device, dtype=torch.float32) / torch.sqrt(var.detach().to(device=device, dtype=torch.float32) + eps)
            w = w * scale.view(-1, 1, 1, 1)
            b = beta.detach().to(device=device, dtype=torch.float32) + (b - mean.detach().to(device=device, dtype=torch.float32)) * scale
        self.stem_superpixel = stem_superpixel
        if stem_superpixel:
            # 6x6 s2 p2 conv over an NHWC4 image == 6x3 s(2,1) p(2,1) conv over (W/2) "super-pixels"
            # of 8 channels (2 pixels x RGB0): tap kx of pixel 2*ox-2+kx -> super-pixel kx//2, parity kx%2.
            assert cin == 3 and kw % 2 == 0
            w4 = torch.zeros(cout, kh, kw // 2, 2, 4, device=device, dtype=torch.float32)
            w4[..., :3] = w.permute(0, 2, 3, 1).reshape(cout, kh, kw // 2, 2, 3)
            wk = w4.reshape(cout, kh * (kw // 2) * 8)
            self.kh, self.kw, self.cin = kh, kw // 2, 8
        else:
            cp = cin if cin_pad is None else cin_pad
            if cp % 8 != 0 or cp < cin:
                raise YmiError(f"conv input channels {cin} (view {cp}) must be padded to a multiple of 8")
            wp = torch.zeros(cout, kh, kw, cp, device=device, dtype=torch.float32)
            wp[..., :cin] = w.permute(0, 2, 3, 1)
            wk = wp.reshape(cout, kh * kw * cp)
            self.kh, self.kw, self.cin = kh, kw, cp
        self.cout = cout
        self.k_real = cin * kh * kw  # algorithmic K (roofline accounting ignores zero padding)
        self.cout_pad = _round_up(cout, 32)
        self.k = wk.shape[1]
        self.k_pad = _round_up(self.k, 32)
        # rows zero-padded to a multiple of 128 (largest cout tile): the pipelined kernel never branches on rows
        packed = torch.zeros(_round_up(cout, 128), self.k_pad, device=device, dtype=torch.float32)
        packed[:cout, : self.k] = wk
        self.w = packed.to(dtype).contiguous()
        self.bias = torch.zeros(self.cout_pad, device=device, dtype=torch.float32)
        self.bias[:cout] = b
        self.dtype = dtype
        self._ktabs: Dict[Tuple[int, int], Tensor] = {}

    def ktab(self, w_in: int, x_cs: int) -> Tensor:
        key = (w_in, x_cs)
        t = self._ktabs.get(key)
        if t is None:
            lib = _lib.load()
            host = (C.c_int32 * (self.k_pad // 8 * 2))()
            check(lib.ymi_conv_build_ktab(self.cin, self.kh, self.kw, w_in, x_cs, self.k_pad, host), "ymi_conv_build_ktab")
            t = torch.tensor(list(host), dtype=torch.int32).to(self.w.device)
            self._ktabs[key] = t
        return t


_TILE_TABLE: Optional[Dict[str, int]] = None
TILE_TABLE_PATH = os.environ.get("YOLORT_AMD_TILE_TABLE_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tiles_gfx950.json")   # override: A/B runs only


def tile_key_str(key: Tuple, dtype: torch.dtype) -> str:
    """stable text form of a conv launch's shape key (the JSON table's key)"""
    n, h, w, cin, cout, kh, kw, s, p, xcs, ycs, odt, res, split, up2, chain = key
    return (f"n{n} {h}x{w} c{cin}->{cout} k{kh}x{kw} s{s[0]}x{s[1]} p{p[0]}x{p[1]} xcs{xcs} ycs{ycs} odt{odt} res{int(bool(res))} split{split} "
            f"up2{int(bool(up2))} chain{int(chain)} {str(dtype).replace('torch.', '')}")


def tile_table() -> Dict[str, int]:
    """the pinned gfx950 tile table (empty when the file is absent: every conv then takes the library heuristic)"""
    global _TILE_TABLE
    if _TILE_TABLE is None:
        import json
        _TILE_TABLE = {}
        if os.path.exists(TILE_TABLE_PATH):
            with open(TILE_TABLE_PATH) as f:
                _TILE_TABLE = {k: int(v) for k, v in json.load(f).get("tiles", {}).items()}
    return _TILE_TABLE


def conv_out_hw(h: int, w: int, k: Tuple[int, int], s: Tuple[int, int], p: Tuple[int, int]) -> Tuple[int, int]:
    return (h + 2 * p[0] - k[0]) // s[0] + 1, (w + 2 * p[1] - k[1]) // s[1] + 1


class Plan:
    """Owns the C plan, the activation buffers and everything the recorded pointers refer to."""

    def __init__(self, device: torch.device, dtype: torch.dtype):
        self.lib = _lib.load(require_gpu=True)
        if device.type != "cuda":
            raise YmiError(f"yolort_amd plans run on an MI355X only (got device {device}); there is no CPU fallback")
        self.device, self.dtype = device, dtype
        self.handle = C.c_void_p(self.lib.ymi_plan_create())
        if not self.handle:
            raise YmiError("ymi_plan_create failed")
        self.keep: List[object] = []      # tensors / descriptors referenced by the C plan
        self.names: List[str] = []
        self.meta: List[dict] = []        # per-op algorithmic flops/bytes for roofline accounting
        self.bytes_allocated = 0
        self.stream: Optional[torch.cuda.Stream] = None
        # zero page for the pipelined conv kernel's out-of-range operand chunks (see yolort_amd.h)
        self.zeros = torch.zeros(1024, device=device, dtype=torch.uint8)
        self.conv_descs: Dict[int, ConvDesc] = {}
        self.io: Dict[int, dict] = {}
        self.chain_1x1 = os.environ.get("YOLORT_AMD_CHAIN", "1") != "0"   # Bottleneck.cv1 chained into C3.cv1+cv2's launch
        # C3.cv3 chained into the last Bottleneck.cv2's launch (ymi_conv_desc.chain_x2): supported and tested, but measured
        # +-0 end to end (the pixel-major producer tiles it needs cost what the saved launch gains) -> off by default
        self.chain_cv3 = os.environ.get("YOLORT_AMD_CHAIN_CV3", "0") == "1"
        # Bottleneck j's 3x3 carries Bottleneck j+1's 1x1 in its epilogue (8-wave halo kernel, 8 x 1 waves: the outputs a wave holds are the 1x1's
        # activation fragments): one launch less per Bottleneck after the first
        # (the same chain at hidden width 128 -- the 40 x 40 C3s of yolov5s -- was measured equal or slower than the two launches, profiles/r04t_setprio_chain128.txt: the chained
        # epilogue reads its 32 KiB of weights per wave from L2 with nothing to overlap; the opt-in was removed in round 5)
        self.chain_next = os.environ.get("YOLORT_AMD_CHAIN_NEXT", "0")   # opt-in: C2 0 ... +2.5 % depending on the box, C5 -0.7 % (profiles/r03u, r03w); "0" off, "1" every hidden width the kernel takes (32 / 64 / 128), "128": only that width
        self.chain_next = False if self.chain_next == "0" else (True if self.chain_next == "1" else int(self.chain_next))
        self.use_v1 = os.environ.get("YOLORT_AMD_CONV_V1", "0") == "1"   # register-staged kernel (debug / A-B)
        # a whole one-Bottleneck C3 of 32 hidden channels in ONE launch (csrc/c3_fused32.hip; yolov5s backbone.body.2).  Default since
        # round 3: bit-identical to the three launches it replaces (tests/test_c3_fused_gpu.py) and 91 vs 196 us on the 160x160 level of
        # the bs-32 yolov5s plan (profiles/r03a_c3fused_ab.txt); YOLORT_AMD_FUSE_C3=0 restores the separate launches
        self.fuse_c3 = os.environ.get("YOLORT_AMD_FUSE_C3", "1") != "0"
        # round 6: C3 blocks of 64 / 128 hidden channels (yolov5s' 80 x 80 and 40 x 40 levels) through the strip kernel (csrc/c3_tile.hip): one launch per block
        # (one Bottleneck) or per Bottleneck (HEAD / MID / TAIL); bit-identical to the separate launches; YOLORT_AMD_C3_TILE=0 restores those
        self.c3_tile_on = os.environ.get("YOLORT_AMD_C3_TILE", "1") != "0"
        # Tile selection is DETERMINISTIC: a pinned per-(shape, dtype) table for gfx950 committed in-tree
        # (yolort_amd/data/tiles_gfx950.json, produced by tools/tune_tiles.py on an MI355X) and, for shapes it does not hold,
        # the library's shape heuristic (tile 0).  Different tiles accumulate K in different orders, so a timing-based choice
        # at plan build would make detections differ between processes / ranks; measuring is therefore opt-in
        # (YOLORT_AMD_AUTOTUNE=1: each conv is timed once per candidate tile on its real buffers; tools/tune_tiles.py
        # writes the winners back into the table).
        self.autotune = os.environ.get("YOLORT_AMD_AUTOTUNE", "0") == "1"
        self.use_tile_table = os.environ.get("YOLORT_AMD_TILE_TABLE", "1") != "0"
        # fp32 PARITY MODE (csrc/conv_f32.hip): fp32 activations between layers and exact fp32 arithmetic, one kernel per
        # reference conv -- no fused pairs / chained convs / folded upsample / fused head, no tile choice.  This is the mode
        # in which the HIP path meets the north-star tolerance against the fp32 CPU reference end to end.
        self.fp32 = dtype == torch.float32
        self.fuse_stem = False   # set_fuse_stem()
        self.res3x3 = {"0": 0, "1": 1}.get(os.environ.get("YOLORT_AMD_RES3X3", "2"), 2)   # tile 132 wherever it fits: 109 -> 86 us at 320^2 (bs 8), 878 -> 572 us for yolov5m's 64 -> 48 (profiles/r03z3_res3x3_*.txt)
        self.rw2 = 0 if os.environ.get("YOLORT_AMD_RW2", "1") == "0" else 1   # tile 134 (conv3x3_rw2.hip) for Conv(64, 128, 3, 2)
        self.rw3 = os.environ.get("YOLORT_AMD_RW3", "0") == "1"   # tile 135 (its K-split form, cin = 128): opt-in until measured
        self.rs = os.environ.get("YOLORT_AMD_RS", "0") == "1"     # tiles 137 / 138 (row-streaming 3x3, conv3x3_rs.hip): opt-in until measured
        if self.fp32:
            # fp32 mode: the LDS-DMA pipelined fp32 tiles (csrc/conv_f32_pipe.hip, tiles 201-206, chosen from the shape by the library) with the fusions that do not
            # change what is summed -- cv1 + cv2 of a C3 in one launch, the PAN's nn.Upsample folded into its producer -- and none that hold 16-bit operands in
            # registers (chained 1x1s, the five-layer C3, the fused stem, the fused head).  YOLORT_AMD_F32_V1=1: the register-staged kernel of rounds 2-4
            # (csrc/conv_f32.hip), one launch per reference conv -- the A/B and bisect partner.
            self.res3x3 = 0
            self.rw2, self.rw3, self.rs = 0, False, False
            self.use_v1 = os.environ.get("YOLORT_AMD_F32_V1", "0") == "1"
            self.chain_1x1, self.chain_cv3 = False, False
            self.fuse_c3 = False
            self.c3_tile_on = False
            self.chain_next = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ymi_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- buffers ----
    def alloc(self, n: int, h: int, w: int, c: int, dtype: Optional[torch.dtype] = None, zero: bool = False) -> View:
        dt = dtype or self.dtype
        numel = n * h * w * c
        # every buffer carries a 256-byte zero tail: the pipelined conv kernel reads out-of-image operand
        # chunks from there (32-bit offset from the view, see ymi_conv_desc.zeros)
        tail = 256 // torch.empty((), dtype=dt).element_size()
        numel = _round_up(max(numel, 1), 8)
        t = (torch.zeros if zero else torch.empty)(numel + tail, device=self.device, dtype=dt)
        if not zero:
            t[numel:].zero_()
        self.keep.append(t)
        self.bytes_allocated += t.numel() * t.element_size()
        v = View(t, 0, n, h, w, c, c)
        v.tail = numel
        return v

    def _record(self, idx: int, name: str, **meta) -> None:
        check(idx, name)
        self.names.append(name)
        self.meta.append(meta)

    # ---- ops ----
    def conv_desc(self, x: View, pc: PackedConv, stride: Tuple[int, int], pad: Tuple[int, int], act: int, y: View, res: Optional[View], tile: int = 0,
                  y2: Optional[View] = None, split: int = 0, up2: bool = False) -> ConvDesc:
        d = ConvDesc()
        d.x, d.w, d.bias = x.ptr, pc.w.data_ptr(), pc.bias.data_ptr()
        is1x1 = pc.kh == 1 and pc.kw == 1 and stride == (1, 1) and pad == (0, 0)
        kt = None if is1x1 else pc.ktab(x.w, x.cs)
        d.ktab = None if kt is None else kt.data_ptr()
        d.y = y.ptr
        d.res = None if res is None else res.ptr
        d.n, d.h, d.w_in, d.cin, d.x_cstride = x.n, x.h, x.w, pc.cin, x.cs
        d.ho, d.wo, d.cout, d.cout_pad, d.y_cstride = y.h, y.w, pc.cout, pc.cout_pad, y.cs
        d.res_cstride = 0 if res is None else res.cs
        d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.k_pad = pc.kh, pc.kw, stride[0], stride[1], pad[0], pad[1], pc.k_pad
        d.act, d.dtype, d.out_dtype, d.tile = act, dtype_code(pc.dtype), dtype_code(y.dtype), tile
        d.y2 = None if y2 is None else y2.ptr
        d.y2_cstride, d.cout_split = (0, 0) if y2 is None else (y2.cs, split)
        d.y2_mode = 1 if (up2 and y2 is not None) else 0
        has_tail = x.tail >= 0
        if y2 is not None and not has_tail:
            raise YmiError("second-output convs need a plan-allocated input (zero tail)")
        v1 = (self.use_v1 and y2 is None) or not has_tail
        d.zeros = None if v1 else x.base.data_ptr() + x.tail * x.base.element_size()
        if v1 and tile >= 0:
            d.tile = -100 if tile == 0 else -tile
        self.keep.extend([pc, kt, d])
        return d

    def conv(self, x: View, pc: PackedConv, stride: int | Tuple[int, int] = 1, pad: int | Tuple[int, int] = 0, act: int = ACT_SILU,
             out: Optional[View] = None, res: Optional[View] = None, out_dtype: Optional[torch.dtype] = None, name: str = "conv", tile: int = 0,
             out2: Optional[View] = None, split: int = 0, up2_out: Optional[View] = None,
             chain: Optional[Tuple] = None) -> View:
        """`out2`/`split`: output channels [split, cout) are written to view `out2` instead of `out`
        (one launch feeding two consumers of the same input, e.g. C3.cv1 + C3.cv2).
        `up2_out`: an (n, 2ho, 2wo, cout) view that additionally receives the whole output nearest-upsampled x2
        (the PAN's nn.Upsample folded into its producer; needs cout % 32 == 0).
        `chain` = (packed 1x1 conv, output view): a second conv on the first `split` (or all) output channels, evaluated in
        this launch's epilogue from registers (Bottleneck.cv1 chained to C3.cv1; K in {32, 64}).  A third element
        `x2` (view) makes it a conv over the concat [fresh outputs | x2] (C3.cv3 chained to the last Bottleneck.cv2)."""
        if act not in (ACT_NONE, ACT_SILU):
            # the legacy r3.1 activations (Hardswish / LeakyReLU(0.1)) are a launch of their own: the convolution runs with YMI_ACT_NONE, ymi_act rewrites its output in
            # place and adds the shortcut AFTER the activation like the reference (common.py:115-116) -- see csrc/preproc_pool.hip act_kernel for why they are not fused
            if out2 is not None or chain is not None or up2_out is not None:
                raise YmiError(f"{name}: second outputs / chained convolutions / the folded upsample carry SiLU or identity only")
            y = self.conv(x, pc, stride, pad, ACT_NONE, out=out, res=None, out_dtype=out_dtype, name=name, tile=tile)
            self.io[self.num_ops - 1]["post_act"] = {"act": act, "res": res}   # (per-launch parity tests run this op and the next as ONE layer)
            return self.act(y, act, res, name=name + ".act")
        s = (stride, stride) if isinstance(stride, int) else tuple(stride)
        p = (pad, pad) if isinstance(pad, int) else tuple(pad)
        x_arg = x
        if pc.stem_superpixel:
            if x.c != 4 or x.cs != 4 or x.w % 2:
                raise YmiError("stem super-pixel conv needs a dense NHWC4 input with even width")
            x = View(x.base, x.off, x.n, x.h, x.w // 2, 8, 8, x.tail)
            s, p = (s[0], 1), (p[0], 1)
        if x.c != pc.cin:
            raise YmiError(f"{name}: input view has {x.c} channels, packed weights expect {pc.cin}")
        ho, wo = conv_out_hw(x.h, x.w, (pc.kh, pc.kw), s, p)
        if out is None:
            assert out2 is None
            cpad = _round_up(pc.cout, 8)
            out = self.alloc(x.n, ho, wo, cpad, out_dtype, zero=cpad != pc.cout).slice_c(0, pc.cout) if cpad != pc.cout else self.alloc(x.n, ho, wo, pc.cout, out_dtype)
        c_first = split if out2 is not None else pc.cout
        if (out.n, out.h, out.w, out.c) != (x.n, ho, wo, c_first):
            raise YmiError(f"{name}: output view {(out.n, out.h, out.w, out.c)} != expected {(x.n, ho, wo, c_first)}")
        if out2 is not None and (out2.n, out2.h, out2.w, out2.c) != (x.n, ho, wo, pc.cout - split):
            raise YmiError(f"{name}: second output view has the wrong shape")
        if up2_out is not None:
            if out2 is not None or pc.cout % 32 or (up2_out.n, up2_out.h, up2_out.w, up2_out.c) != (x.n, 2 * ho, 2 * wo, pc.cout) or out.dtype != self.dtype:
                raise YmiError(f"{name}: the upsampled second output needs cout % 32 == 0, an (n, 2ho, 2wo, cout) view and no channel split")
            d = self.conv_desc(x, pc, s, p, act, out, res, tile, up2_out, 0, up2=True)
        else:
            d = self.conv_desc(x, pc, s, p, act, out, res, tile, out2, split)
        if chain is not None:
            pc2, tv = chain[0], chain[1]
            x2 = chain[2] if len(chain) > 2 else None
            k1 = split if out2 is not None else pc.cout
            k2 = 0 if x2 is None else x2.c
            if (pc2.kh != 1 or pc2.kw != 1 or pc2.cin != k1 + k2 or pc2.k_pad != k1 + k2 or (tv.n, tv.h, tv.w, tv.c) != (x.n, ho, wo, pc2.cout) or up2_out is not None
                    or (x2 is not None and ((x2.n, x2.h, x2.w) != (x.n, ho, wo) or out2 is not None or k2 % 16))):
                raise YmiError(f"{name}: chained conv must be a 1x1 over the first {k1} output channels (+ the second source's) with matching views")
            d.chain_w, d.chain_bias, d.chain_y = pc2.w.data_ptr(), pc2.bias.data_ptr(), tv.ptr
            d.chain_cout, d.chain_y_cstride = pc2.cout, tv.cs
            if x2 is not None:
                d.chain_x2, d.chain_x2_cstride, d.chain_k2 = x2.ptr, x2.cs, k2
            self.keep.append(pc2)
        self.conv_descs[self.num_ops] = d   # op index -> descriptor (the fused stem path re-issues op 0 from planar images)
        # op index -> the views this launch reads / writes (per-layer parity tests fill `x` / `res` and read the outputs)
        self.io[self.num_ops] = {"name": name, "x": x_arg, "y": out, "y2": out2, "split": split if out2 is not None else 0, "up2": up2_out, "res": res,
                                 "chain_y": None if chain is None else chain[1], "chain_x2": None if chain is None or len(chain) < 3 else chain[2],
                                 "stride": s, "pad": p}
        if tile == 0 and d.zeros:
            tkey = (x.n, x.h, x.w, pc.cin, pc.cout, pc.kh, pc.kw, s, p, x.cs, out.cs, dtype_code(out.dtype), res is not None, split, up2_out is not None,
                    # chain kind: 0 none, 1 chained 1x1 over this launch's output, 2 + the second source's channels (cv3 over the concat) -- the
                    # candidate tiles differ per kind, so a table entry must not be applied across kinds
                    0 if chain is None else (1 if len(chain) < 3 or chain[2] is None else 2 + chain[2].c))
            pinned = tile_table().get(tile_key_str(tkey, self.dtype), 0) if self.use_tile_table else 0
            if self.autotune:
                d.tile = self._autotune_tile(d, tkey, chain)
            elif pinned >= 132 and os.environ.get("YOLORT_AMD_RULES_FIRST", "0") != "1" and self._pinned_ok(d, pinned):
                d.tile = pinned   # an entry written by a tuner that knew the resident-weights kernels (tiles 132 ...): it has measured them against each other on this shape
            elif self.rs and self._rs_ok(d):
                d.tile = 136 + d.sh   # row-streaming 3x3 (conv3x3_rs.hip): tile 137 (stride 1, 64 -> 64, no shortcut) / 138 (stride 2, 64 -> 128); YOLORT_AMD_RS=0 keeps tiles 133 / 134
            elif self.res3x3 and self._res3x3_ok(d):
                # resident-weights persistent 3x3 (conv3x3_res.hip): ahead of the table on every layer it fits (same-box A/B); its
                # register-weights variant (conv3x3_rw.hip, bit-identical, 9-10 % faster: profiles/r03z14_rw3x3.txt) where that one fits; YOLORT_AMD_RES3X3=1: tile 132 only
                d.tile = 133 if (self.res3x3 == 2 and d.cout == 64 and d.act == ACT_SILU and not d.chain_w) else 132
            elif self.rw2 and self._rw2_ok(d):
                d.tile = 134   # stride-2 register-weights 3x3 (conv3x3_rw2.hip): ahead of the table for Conv(64, 128, 3, 2); YOLORT_AMD_RW2=0 keeps the table's tile
            elif self.rw3 and self._rw3_ok(d):
                d.tile = 135   # ... its K-split form for Conv(128, 128 / 256, 3, 2); YOLORT_AMD_RW3=0 keeps the table's tile
            elif self.use_tile_table:
                # a table entry >= 132 that THIS launch does not meet the preconditions of (another activation, a chained conv, an opted-out kernel: _pinned_ok said no above)
                # must not come back through the fallback: the library's shape heuristic (tile 0) takes the launch
                d.tile = pinned if (pinned < 132 or self._pinned_ok(d, pinned)) else 0
        esz = 4 if self.fp32 else 2
        if self.fp32 and d.tile == 0 and d.zeros:
            d.tile = int(self.lib.ymi_conv_f32_pick_tile(x.n * ho * wo, pc.cout_pad))   # what the library would choose itself: recorded so that the layer tables name the tile
        flops = 2.0 * x.n * ho * wo * pc.cout * pc.k_real  # algorithmic MACs (zero padding not counted)
        # algorithmic bytes follow SURVEY.md 8d: every reference conv reads its input once and writes its
        # output once; a fused cv1+cv2 launch stands for two reference convs, so its input counts twice.
        ref_reads = 2 if out2 is not None else 1
        cin_real = pc.k_real // (pc.kh * pc.kw)   # the reference conv's input channels: zero-padded views (yolov5m's 48 -> 64 hidden width, the RGB0 super-pixels) do not count
        chain_flops = chain_bytes = 0.0
        if chain is not None:   # the chained conv is a reference conv of its own: reads its input once, writes its output once
            chain_flops = 2.0 * x.n * ho * wo * chain[0].cout * chain[0].k_real
            chain_bytes = float(x.n * ho * wo * (chain[0].k_real + chain[0].cout) * esz + chain[0].cout * chain[0].k_real * esz)
        self._record(self.lib.ymi_plan_add_conv(self.handle, C.byref(d)), name, kind="conv",
                     flops=flops + chain_flops,
                     bytes=float(ref_reads * x.n * x.h * x.w * cin_real * esz + x.n * ho * wo * pc.cout * out.base.element_size() + pc.cout * pc.k_real * esz) + chain_bytes,
                     ref_convs=ref_reads + (1 if chain is not None else 0), tile=int(d.tile),
                     shape=f"{x.c}->{pc.cout} k{pc.kh}x{pc.kw} s{s[0]} {x.h}x{x.w}->{ho}x{wo}")
        return out

    def _pinned_ok(self, d: ConvDesc, tile: int) -> bool:
        """A table entry >= 132 names a kernel with hard preconditions the table key does not carry (activation, a chained conv, the shortcut's stride, 32-bit offsets):
        it is taken only when THIS launch meets them, and an explicit opt-out (YOLORT_AMD_RES3X3=0 / RW2=0 / RW3=0 / RS=0) wins over the table.  Otherwise the rule chain /
        the general tiles apply -- a Conv with another activation on a pinned shape builds its plan instead of failing."""
        env = os.environ.get
        if tile == 132:
            return bool(self.res3x3) and self._res3x3_ok(d)
        if tile == 133:
            return self.res3x3 == 2 and self._res3x3_ok(d) and d.cout == 64 and d.act == ACT_SILU and not d.chain_w
        if tile == 134:
            return bool(self.rw2) and self._rw2_ok(d)
        if tile == 135:
            return env("YOLORT_AMD_RW3", "1") != "0" and self._rw3_ok(d)
        if tile in (137, 138):
            return env("YOLORT_AMD_RS", "1") != "0" and self._rs_ok(d) and tile == 136 + d.sh
        return tile in (141, 142, 143, 144, 145, 151, 152, 155) and d.out_dtype == d.dtype   # row-transposed-store forms: general tiles (the kernels fall back inside for the cases they do not take)

    @staticmethod
    def _res3x3_ok(d: ConvDesc) -> bool:
        chain_ok = (not d.chain_w) or (not d.chain_x2 and d.cout_pad == d.cout and d.cout in (32, 64))
        # the launchers' 32-bit offset conditions (conv3x3_rw.hip / conv3x3_res.hip): a concat-slice output has the stride of the whole buffer
        if (d.n * d.ho * d.wo + 1) * max(d.y_cstride, d.res_cstride) >= 2 ** 31 or d.n * d.h * d.w_in * d.x_cstride >= 2 ** 31:
            return False
        return (d.kh == 3 and d.kw == 3 and d.sh == 1 and d.sw == 1 and d.ph == 1 and d.pw == 1 and d.cin in (48, 64) and d.k_pad >= 9 * d.cin and d.cout <= 64
                and d.cout_split == 0 and d.y2_mode == 0 and d.out_dtype == d.dtype and bool(d.zeros) and chain_ok)

    @staticmethod
    def _rw2_ok(d: ConvDesc) -> bool:
        return (d.kh == 3 and d.kw == 3 and d.sh == 2 and d.sw == 2 and d.ph == 1 and d.pw == 1 and d.cin == 64 and d.k_pad >= 576 and d.cout == 128 and d.cout_pad >= 128
                and d.cout_split == 0 and d.y2_mode == 0 and d.out_dtype == d.dtype and bool(d.zeros) and not d.chain_w and not d.res and d.act == ACT_SILU
                and (d.n * d.ho * d.wo + 1) * d.y_cstride < 2 ** 31 and d.n * d.h * d.w_in * d.x_cstride < 2 ** 31)

    @staticmethod
    def _rs_ok(d: ConvDesc) -> bool:
        return (d.kh == 3 and d.kw == 3 and d.sh == d.sw and d.sh in (1, 2) and d.ph == 1 and d.pw == 1 and d.cin == 64 and d.k_pad >= 576 and d.cout == (64 if d.sh == 1 else 128)
                and d.cout_pad >= d.cout and d.cout_split == 0 and d.y2_mode == 0 and d.out_dtype == d.dtype and bool(d.zeros) and not d.chain_w and not d.res and d.act == ACT_SILU
                and (d.n * d.ho * d.wo + 1) * d.y_cstride < 2 ** 31 and d.n * d.h * d.w_in * d.x_cstride < 2 ** 31)

    @staticmethod
    def _rw3_ok(d: ConvDesc) -> bool:
        return (d.kh == 3 and d.kw == 3 and d.sh == 2 and d.sw == 2 and d.ph == 1 and d.pw == 1 and d.cin == 128 and d.k_pad >= 1152 and d.cout in (128, 256) and d.cout_pad >= d.cout
                and d.cout_split == 0 and d.y2_mode == 0 and d.out_dtype == d.dtype and bool(d.zeros) and not d.chain_w and not d.res and d.act == ACT_SILU
                and (d.n * d.ho * d.wo + 1) * d.y_cstride < 2 ** 31 and d.n * d.h * d.w_in * d.x_cstride < 2 ** 31)

    _TUNE_CACHE: Dict[Tuple, int] = {}
    _TUNE_TIMES: Dict[str, Dict[str, float]] = {}

    def _autotune_tile(self, d: ConvDesc, key: Tuple, chain=None) -> int:
        key = key + (self.dtype,)
        hit = Plan._TUNE_CACHE.get(key)
        if hit is not None:
            return hit
        if self.fp32:
            return self._autotune_time(d, key, [int(t) for t in os.environ.get("YOLORT_AMD_F32_TUNE_TILES", "201,202,203,204,205,206").split(",")])
        cands = [11, 12, 14, 15, 21, 22, 24, 25, 27]
        if d.cout_pad <= 32:
            cands = [13, 23, 26, 25]
        elif d.cout_pad <= 64:
            cands = [12, 15, 22, 25, 23, 26, 27]
        if d.cin % 32 == 0 and d.kh * d.kw <= 32:
            cands = cands + [t + 50 for t in cands]   # the same tiles with the software-pipelined main loop (61..65, 71..77)
            cands = cands + ([69, 70] if d.cout_pad > 32 else []) + ([66, 68] if d.cout_pad >= 128 else [])   # 3-stage / 256-pixel tiles
        if os.environ.get("YOLORT_AMD_TUNE_TP", "1") != "0" and d.out_dtype == d.dtype and d.y2_mode in (0, 1) and chain is None and d.cin % 32 == 0 and d.kh * d.kw <= 32 and \
                d.k_pad == d.kh * d.kw * d.cin:   # (y2_mode 1, round 4: the row-transposed stores also write the upsampled copy, as whole rows)
            # row-transposed-store forms of tiles 12 / 21 / 66 / 61 / 71 and of the 8-wave implicit GEMM (141-145, 151-155): bit-identical
            # to their base tiles (tests/test_c3_fused_gpu.py, tests/test_hipsim_kernels.py), offered to the tuner since round 3
            cands = cands + [t for t, base in ((141, 12), (142, 21), (143, 66), (144, 61), (145, 71)) if base in cands or d.y2_mode == 1]
            cands = cands + ([155] if d.cout_pad > 128 else []) + ([151] if d.cout_pad > 64 else []) + ([152] if 32 < d.cout_pad <= 128 else [])
        if d.kh == 1 and d.kw == 1 and d.sh == 1 and d.sw == 1 and d.cin % 32 == 0 and d.cin <= 128 and d.k_pad == d.cin and d.out_dtype == d.dtype and d.y2_mode == 0 and d.cout % 32 == 0:
            # streaming 1x1 kernel (conv1x1_stream.hip): variant = cout tiles of 32 per wave; a split / chained conv fixes the block width
            k1 = d.cout_split if d.cout_split > 0 else 0
            if chain is not None:
                cands = cands + [120 + (k1 if k1 else d.cout) // 32]
            else:
                cands = cands + [120 + t for t in (1, 2, 3, 4) if (k1 == 0 or k1 % (32 * t) == 0)]
        if d.cin % 32 == 0 and d.kh * d.kw <= 32 and d.k_pad == d.kh * d.kw * d.cin:
            # 8-wave implicit GEMM with 64-deep steps (conv_igemm8.hip); a chained 1x1 needs pixel-major waves of its K1 width
            if chain is not None:
                k1 = d.cout_split if d.cout_split > 0 else d.cout
                cands = cands + ([114] if k1 == 32 else ([113] if k1 == 64 else ([116] if k1 == 128 else [])))
            else:
                cands = cands + ([114] if d.cout_pad <= 32 else ([112, 113] if d.cout_pad <= 64 else ([111, 112, 116] if d.cout_pad <= 128 else [111, 115, 112])))
                if d.cout % 192 == 0 and d.cout_split == 0:
                    cands = cands + [120]   # ... in 192-cout blocks (yolov5m's 192 / 384 / 768-cout layers)
                # (tiles 117-119 = 111 / 116 / 112 with a three-deep stage ring: measured equal to the two-deep ring on every
                # yolov5s layer, profiles/r02z_conv_bench_ring3.txt -- the steps are not DMA-latency-bound; not offered to the tuner)
        if d.kh == 3 and d.kw == 3 and d.sh == 1 and d.sw == 1 and d.ph == 1 and d.pw == 1 and d.cin % 32 == 0 and d.cout_split == 0 and d.k_pad == 9 * d.cin:
            # LDS-halo kernel variants (activation patch resident in LDS across the nine taps)
            cands = cands + ([33, 36] if d.cout_pad <= 32 else ([32, 35, 37, 33] if d.cout_pad <= 64 else [31, 34, 32, 37]))
            if chain is None:   # 8-wave halo kernel (conv_halo8.hip): 256-pixel patches, <= 2 DMA pieces per wave per step
                cands = cands + ([94] if d.cout_pad <= 32 else ([92, 93] if d.cout_pad <= 64 else [91, 92, 93, 95]))
                if d.cout % 96 == 0:
                    cands = cands + [96]   # ... in 96-cout blocks (yolov5m's 96 / 192-cout layers)
            else:               # ... its 8 x 1 forms take a chained 1x1 whose K is the whole cout width (round 3)
                k1 = d.cout_split if d.cout_split > 0 else d.cout
                cands = cands + ([94] if k1 == 32 == d.cout_pad else ([93] if k1 == 64 == d.cout_pad else ([95] if k1 == 128 == d.cout_pad else [])))
        if d.kh == 3 and d.kw == 3 and d.sh == d.sw and d.sh in (1, 2) and d.ph == 1 and d.pw == 1 and d.cin == 32 and d.cout in (32, 64) and d.k_pad == 288 and \
                d.out_dtype == d.dtype and d.y2_mode != 2 and chain is None:
            cands = cands + [131]   # resident-weights persistent 3x3 (conv3x3_c32.hip)
        if self._rw2_ok(d):
            cands = cands + [134]   # stride-2 register-weights 3x3 (conv3x3_rw2.hip)
        if self._rw3_ok(d):
            cands = cands + [135]   # ... K split over two waves, cin = 128
        if self._rs_ok(d):
            cands = cands + [136 + d.sh]   # row-streaming 3x3 (conv3x3_rs.hip)
        if self._res3x3_ok(d):
            cands = cands + [132]   # ... cin = 48 / 64, stride 1, cross-tile patch prefetch (conv3x3_res.hip)
            if d.cout == 64 and d.act == ACT_SILU and not d.chain_w:
                cands = cands + [133]   # ... weights in registers, two blocks per CU (conv3x3_rw.hip)
        if d.cin == 8 and d.kh == 6 and d.kw == 3 and d.sh == 2 and d.sw == 1 and d.x_cstride == 8 and d.cout_pad <= 64 and not d.res:
            cands = cands + [41]   # dedicated stem kernel
        return self._autotune_time(d, key, cands)

    def _autotune_time(self, d: ConvDesc, key: Tuple, cands: List[int]) -> int:
        best, best_ms = 0, float("inf")
        stream = _lib.stream_ptr()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ok = []
        for t in cands:
            d.tile = t
            if self.lib.ymi_conv2d(C.byref(d), stream) >= 0:   # else: configuration not applicable
                ok.append(t)
        torch.cuda.synchronize()
        times = {t: float("inf") for t in ok}
        for _round in range(3):           # interleaved rounds, best-of: robust against clock / neighbour noise
            for t in ok:
                d.tile = t
                reps = 4
                ev[0].record()
                for _ in range(reps):
                    self.lib.ymi_conv2d(C.byref(d), stream)
                ev[1].record()
                ev[1].synchronize()
                times[t] = min(times[t], ev[0].elapsed_time(ev[1]) / reps)
        for t in ok:
            if times[t] < best_ms:
                best, best_ms = t, times[t]
        Plan._TUNE_CACHE[key] = best
        Plan._TUNE_TIMES[tile_key_str(key[:-1], key[-1])] = {str(t): round(times[t] * 1e3, 2) for t in ok}   # us per candidate (tools/tune_tiles.py)
        return best

    def c3_fused(self, x: View, pc12: PackedConv, pcm1: PackedConv, pcm2: PackedConv, pc3: PackedConv, out: Optional[View] = None,
                 name: str = "c3.fused") -> View:
        """cv3(cat(x1 + m.cv2(m.cv1(x1)), cv2(x))), x1 = cv1(x), in ONE launch (ymi_c3_fused; reference common.py:172-173 with
        :115-116 inlined).  `pc12` = cv1 and cv2 stacked along cout (C3.packed_pair).  Bit-identical to the separate launches."""
        c_ = pcm1.cout
        if out is None:
            out = self.alloc(x.n, x.h, x.w, pc3.cout)
        if ((pc12.kh, pc12.kw, pcm1.kh, pcm1.kw, pcm2.kh, pcm2.kw, pc3.kh, pc3.kw) != (1, 1, 1, 1, 3, 3, 1, 1) or pc12.cin != x.c or pc12.cout != 2 * c_
                or pcm1.cin != c_ or pcm2.cin != c_ or pcm2.cout != c_ or pc3.cin != 2 * c_ or (out.n, out.h, out.w, out.c) != (x.n, x.h, x.w, pc3.cout)
                or out.dtype != self.dtype or x.dtype != self.dtype):
            raise YmiError(f"{name}: the packed convolutions do not form a one-Bottleneck C3 over the given views")
        d = C3Desc()
        d.x, d.y = x.ptr, out.ptr
        d.w12, d.b12, d.wm1, d.bm1 = pc12.w.data_ptr(), pc12.bias.data_ptr(), pcm1.w.data_ptr(), pcm1.bias.data_ptr()
        d.wm2, d.bm2, d.w3, d.b3 = pcm2.w.data_ptr(), pcm2.bias.data_ptr(), pc3.w.data_ptr(), pc3.bias.data_ptr()
        d.n, d.h, d.w, d.x_cstride, d.y_cstride, d.dtype = x.n, x.h, x.w, x.cs, out.cs, dtype_code(self.dtype)
        d.c_in, d.c_hidden, d.c_out, d.n_bottlenecks, d.shortcut = x.c, c_, pc3.cout, 1, 1
        d.k12_pad, d.km1_pad, d.km2_pad, d.k3_pad = pc12.k_pad, pcm1.k_pad, pcm2.k_pad, pc3.k_pad
        self.keep.extend([pc12, pcm1, pcm2, pc3, d])
        npix, esz = x.n * x.h * x.w, 2
        convs = [(x.c, 2 * c_, 1), (c_, c_, 1), (c_, c_, 9), (2 * c_, pc3.cout, 1)]   # cv1 + cv2 (two reference convs reading x), m.cv1, m.cv2, cv3
        flops = sum(2.0 * npix * ci * co * k for ci, co, k in convs)
        # algorithmic bytes as SURVEY.md 8d counts them: every reference conv reads its input once and writes its output once
        nbytes = float(npix * esz * (2 * x.c + 2 * c_ + 2 * c_ + 2 * c_ + 2 * c_ + pc3.cout) + esz * sum(ci * co * k for ci, co, k in convs))
        self.io[self.num_ops] = {"name": name, "x": x, "y": out, "y2": None, "split": 0, "up2": None, "res": None, "chain_y": None, "chain_x2": None,
                                 "stride": (1, 1), "pad": (0, 0), "fused_c3": True}
        self._record(self.lib.ymi_plan_add_c3_fused(self.handle, C.byref(d)), name, kind="conv", flops=flops, bytes=nbytes, ref_convs=5, tile=-1,
                     shape=f"C3 {x.c}->{pc3.cout} hidden {c_} n1 {x.h}x{x.w}")
        return out

    def c3_tile_ok(self, x: View, c_: int) -> bool:
        """does the strip kernel (csrc/c3_tile.hip) hold a geometry for a C3 of hidden width c_ over x?"""
        if self.fp32 or self.use_v1 or not self.c3_tile_on or c_ not in (64, 128) or x.c % 32 or x.cs % 8 or x.dtype != self.dtype:
            return False
        d = C3Desc()
        d.n, d.h, d.w, d.c_hidden, d.c_out = x.n, x.h, x.w, c_, 2 * c_
        return bool(self.lib.ymi_c3_tile_supported(C.byref(d)))

    def c3_tile(self, mode: int, pcm1: PackedConv, pcm2: PackedConv, shortcut: bool, x: Optional[View] = None, pc12: Optional[PackedConv] = None,
                pc3: Optional[PackedConv] = None, out: Optional[View] = None, y1_in: Optional[View] = None, y1_out: Optional[View] = None,
                y2: Optional[View] = None, name: str = "c3.tile") -> Optional[View]:
        """One launch of the strip kernel (ymi_c3_fused, instance (2) of include/yolort_amd.h; reference common.py:172-173 with :115-116 inlined):
        mode 0 the whole one-Bottleneck block x -> out; 1 HEAD x -> y1_out, y2; 2 MID y1_in -> y1_out; 3 TAIL y1_in, y2 -> out.
        Bit-identical to the separate launches (1x1: k ascending; 3x3: the LDS-halo kernels' order)."""
        c_ = pcm1.cout
        has_a, has_d = mode in (0, 1), mode in (0, 3)
        src = x if has_a else y1_in
        if src is None or (has_a and pc12 is None) or (has_d and pc3 is None):
            raise YmiError(f"{name}: mode {mode} is missing an operand")
        if has_d and out is None:
            out = self.alloc(src.n, src.h, src.w, pc3.cout)
        if ((pcm1.kh, pcm1.kw, pcm2.kh, pcm2.kw) != (1, 1, 3, 3) or pcm1.cin != c_ or pcm2.cin != c_ or pcm2.cout != c_
                or (has_a and ((pc12.kh, pc12.kw) != (1, 1) or pc12.cin != x.c or pc12.cout != 2 * c_))
                or (has_d and ((pc3.kh, pc3.kw) != (1, 1) or pc3.cin != 2 * c_ or pc3.cout != 2 * c_ or (out.n, out.h, out.w, out.c) != (src.n, src.h, src.w, 2 * c_)))
                or (not has_a and y1_in.c != c_) or (not has_d and (y1_out is None or (y1_out.n, y1_out.h, y1_out.w, y1_out.c) != (src.n, src.h, src.w, c_)))
                or (mode in (1, 3) and (y2 is None or (y2.n, y2.h, y2.w, y2.c) != (src.n, src.h, src.w, c_)))):
            raise YmiError(f"{name}: the packed convolutions / views do not form mode {mode} of a C3 with {c_} hidden channels")
        d = C3Desc()
        d.n, d.h, d.w, d.dtype = src.n, src.h, src.w, dtype_code(self.dtype)
        d.c_in, d.c_hidden, d.c_out, d.n_bottlenecks, d.shortcut, d.mode = (x.c if has_a else c_), c_, 2 * c_, 1, 1 if shortcut else 0, mode
        if has_a:
            d.x, d.x_cstride = x.ptr, x.cs
            d.w12, d.b12, d.k12_pad = pc12.w.data_ptr(), pc12.bias.data_ptr(), pc12.k_pad
        else:
            d.y1_in, d.y1_in_cstride = y1_in.ptr, y1_in.cs
        d.wm1, d.bm1, d.km1_pad = pcm1.w.data_ptr(), pcm1.bias.data_ptr(), pcm1.k_pad
        d.wm2, d.bm2, d.km2_pad = pcm2.w.data_ptr(), pcm2.bias.data_ptr(), pcm2.k_pad
        if has_d:
            d.y, d.y_cstride = out.ptr, out.cs
            d.w3, d.b3, d.k3_pad = pc3.w.data_ptr(), pc3.bias.data_ptr(), pc3.k_pad
        else:
            d.y1_out, d.y1_out_cstride = y1_out.ptr, y1_out.cs
        if mode in (1, 3):
            d.y2, d.y2_cstride = y2.ptr, y2.cs
        nb = int(self.lib.ymi_c3_blob_bytes(C.byref(d)))
        if nb <= 0:
            raise YmiError(f"{name}: no weight stream layout for this descriptor")
        blob = torch.empty(nb, device=self.device, dtype=torch.uint8)
        check(self.lib.ymi_c3_pack(C.byref(d), blob.data_ptr(), _lib.stream_ptr() if self.device.type == "cuda" else None), "ymi_c3_pack")
        d.wblob = blob.data_ptr()
        if not hasattr(self, "_const_tensors"):
            self._const_tensors = set()
        self._const_tensors.add(id(blob))   # (Plan.export: a constant region, its contents are part of the file)
        self.keep.extend([pc12, pcm1, pcm2, pc3, blob, d])
        npix, esz = src.n * src.h * src.w, 2
        # the reference convolutions this launch stands for: (cin, cout, taps); every one reads its input once and writes its output once
        convs = ([(x.c, c_, 1), (x.c, c_, 1)] if has_a else []) + [(c_, c_, 1), (c_, c_, 9)] + ([(2 * c_, 2 * c_, 1)] if has_d else [])
        flops = sum(2.0 * npix * ci * co * k for ci, co, k in convs)
        nbytes = float(sum(npix * esz * (ci + co) + esz * ci * co * k for ci, co, k in convs))
        # the launch's roofline bound stays the sum of its reference layers' own bounds (one bound per reference conv), not the bound of the summed launch:
        # `layers` = (flops, bytes) of each reference conv this launch stands for
        layers = [(2.0 * npix * ci * co * k, float(npix * esz * (ci + co) + esz * ci * co * k)) for ci, co, k in convs]
        self.io[self.num_ops] = {"name": name, "x": x, "y": out, "y2": y2, "split": 0, "up2": None, "res": None, "chain_y": None, "chain_x2": None,
                                 "stride": (1, 1), "pad": (0, 0), "fused_c3": True, "c3_mode": mode, "y1_in": y1_in, "y1_out": y1_out, "shortcut": bool(shortcut)}
        what = {0: "C3", 1: "C3 head", 2: "Bottleneck", 3: "C3 tail"}[mode]
        self._record(self.lib.ymi_plan_add_c3_fused(self.handle, C.byref(d)), name, kind="conv", flops=flops, bytes=nbytes, ref_convs=len(convs), tile=-3, layers=layers,
                     shape=f"{what} {(x.c if has_a else c_)}->{2 * c_ if has_d else c_} hidden {c_} {src.h}x{src.w}")
        return out if has_d else y1_out

    def spp_pool(self, buf: View, c: int, name: str = "spp_pool") -> None:
        assert buf.c == 4 * c
        self._record(self.lib.ymi_plan_add_spp_pool(self.handle, buf.ptr, buf.n, buf.h, buf.w, c, buf.cs, dtype_code(buf.dtype)), name,
                     kind="pool", flops=0.0, bytes=float(buf.n * buf.h * buf.w * c * buf.base.element_size() * 4), shape=f"c{c} {buf.h}x{buf.w}")

    def upsample2x(self, x: View, out: View, name: str = "upsample2x") -> View:
        assert (out.n, out.h, out.w, out.c) == (x.n, 2 * x.h, 2 * x.w, x.c)
        self._record(self.lib.ymi_plan_add_upsample2x(self.handle, x.ptr, x.cs, x.n, x.h, x.w, x.c, out.ptr, out.cs, dtype_code(x.dtype)), name,
                     kind="upsample", flops=0.0, bytes=float(x.n * x.h * x.w * x.c * x.base.element_size() * 5), shape=f"c{x.c} {x.h}x{x.w}")
        return out

    def act(self, y: View, act: int, res: Optional[View] = None, name: str = "act") -> View:
        """y <- act(y) (+ res) in place (ymi_act: the legacy r3.1 activations after a convolution run with YMI_ACT_NONE)"""
        if res is not None and (res.n, res.h, res.w, res.c) != (y.n, y.h, y.w, y.c):
            raise YmiError(f"{name}: shortcut view does not match the output")
        esz = y.base.element_size()
        self._record(self.lib.ymi_plan_add_act(self.handle, y.ptr, y.cs, y.n * y.h * y.w, y.c, dtype_code(y.dtype), act, None if res is None else res.ptr, 0 if res is None else res.cs),
                     name, kind="act", flops=0.0, bytes=float(y.n * y.h * y.w * y.c * esz * (3 if res is not None else 2)), shape=f"c{y.c} {y.h}x{y.w}")
        return y

    def copy(self, x: View, out: View, name: str = "copy") -> View:
        assert (out.n, out.h, out.w, out.c) == (x.n, x.h, x.w, x.c)
        self._record(self.lib.ymi_plan_add_copy_view(self.handle, x.ptr, x.cs, x.n * x.h * x.w, x.c, out.ptr, out.cs, dtype_code(x.dtype)), name,
                     kind="copy", flops=0.0, bytes=float(x.n * x.h * x.w * x.c * x.base.element_size() * 2), shape=f"c{x.c} {x.h}x{x.w}")
        return out

    def post_desc(self, levels: Sequence[Tuple[int, int]], n: int, strides: Sequence[float], anchors: Sequence[Sequence[float]], num_classes: int,
                  score_thresh: float, nms_thresh: float, detections_per_img: int, cand_cap: int, rescale: Optional[Tensor] = None,
                  logits: Optional[Sequence[View]] = None, flags: int = 0) -> Tuple["PostBuffers", PostDesc]:
        """descriptor + output slab + workspace of one post-process; `levels` = [(h, w)] per pyramid level"""
        total_anchors = sum(3 * h * w for h, w in levels)
        pb = PostBuffers(self, n, detections_per_img, total_anchors, cand_cap, rescale)
        d = PostDesc()
        for i, (h, w) in enumerate(levels):
            d.lh[i], d.lw[i] = h, w
            d.stride[i] = float(strides[i])
            for k in range(6):
                d.anchors[i][k] = float(anchors[i][k])
            if logits is not None:
                v = logits[i]
                if v.dtype != torch.float32:
                    raise YmiError("post-process expects fp32 head logits")
                d.logits[i], d.lcstride[i] = v.ptr, v.cs
        d.num_levels, d.n, d.num_classes = len(levels), n, num_classes
        d.score_thresh, d.nms_thresh, d.detections_per_img = score_thresh, nms_thresh, detections_per_img
        d.rescale = None if pb.rescale is None else pb.rescale.data_ptr()
        d.out_boxes, d.out_scores, d.out_labels, d.out_count = pb.boxes.data_ptr(), pb.scores.data_ptr(), pb.labels.data_ptr(), pb.count.data_ptr()
        d.status = pb.status.data_ptr()
        d.out_slab = pb.slab.data_ptr()
        d.ws, d.ws_bytes, d.cand_cap = pb.ws.data_ptr(), pb.ws.numel(), cand_cap
        d.flags = flags
        pb.total_anchors = total_anchors
        self.keep.extend([pb, d])
        return pb, d

    def postprocess(self, logits: Sequence[View], strides: Sequence[float], anchors: Sequence[Sequence[float]], num_classes: int,
                    score_thresh: float, nms_thresh: float, detections_per_img: int, cand_cap: int, rescale: Optional[Tensor] = None,
                    flags: int = 0) -> "PostBuffers":
        """decode of stored fp32 logits + sort + NMS + top-k as ONE op (the unfused form)"""
        pb, d = self.post_desc([(v.h, v.w) for v in logits], logits[0].n, strides, anchors, num_classes, score_thresh, nms_thresh, detections_per_img,
                               cand_cap, rescale, logits=logits, flags=flags)
        self._record(self.lib.ymi_plan_add_postprocess(self.handle, C.byref(d)), "postprocess", kind="post", flops=0.0,
                     bytes=float(sum(v.n * v.h * v.w * 3 * (num_classes + 5) * 4 for v in logits)), shape=f"A={pb.total_anchors}")
        return pb

    # fused head: post_begin -> head_decode per level -> post_finish (the logits never reach memory)
    def post_begin(self, d: PostDesc) -> None:
        self._record(self.lib.ymi_plan_add_post_begin(self.handle, C.byref(d)), "post_begin", kind="post", flops=0.0, bytes=0.0, shape="counters")

    def head_decode(self, x: View, pc: PackedConv, d: PostDesc, level: int, name: str = "head") -> None:
        if x.c != pc.cin:
            raise YmiError(f"{name}: input view has {x.c} channels, packed weights expect {pc.cin}")
        if x.tail < 0:
            raise YmiError(f"{name}: the fused head needs a plan-allocated input (zero tail)")
        cd = self.conv_desc(x, pc, (1, 1), (0, 0), ACT_NONE, x, None)   # y is ignored by the fused head
        cd.y = None
        cd.y_cstride, cd.out_dtype = 0, dtype_code(torch.float32)
        k_real = int(d.num_classes) + 5
        self._record(self.lib.ymi_plan_add_head_decode(self.handle, C.byref(cd), C.byref(d), level), name, kind="conv",
                     flops=2.0 * x.n * x.h * x.w * 3 * k_real * pc.k_real,
                     bytes=float(x.n * x.h * x.w * x.c * 2 + 3 * k_real * pc.k * 2 + x.n * x.h * x.w * 3 * 16),
                     ref_convs=1, tile=0, shape=f"{x.c}->{3 * k_real} k1x1 s1 {x.h}x{x.w} +decode")

    def head_decode_group(self, xs: Sequence[View], pcs: Sequence[PackedConv], d: PostDesc, name: str = "head") -> None:
        """the fused heads of all pyramid levels in ONE launch (ymi_conv_head_decode_group)"""
        arr = (ConvDesc * len(xs))()
        flops = nbytes = 0.0
        k_real = int(d.num_classes) + 5
        for i, (x, pc) in enumerate(zip(xs, pcs)):
            if x.c != pc.cin or x.tail < 0:
                raise YmiError(f"{name}: level {i} needs a plan-allocated input view matching the packed weights")
            cd = self.conv_desc(x, pc, (1, 1), (0, 0), ACT_NONE, x, None)
            cd.y = None
            cd.y_cstride, cd.out_dtype = 0, dtype_code(torch.float32)
            C.memmove(C.byref(arr, i * C.sizeof(ConvDesc)), C.byref(cd), C.sizeof(ConvDesc))
            flops += 2.0 * x.n * x.h * x.w * 3 * k_real * pc.k_real
            nbytes += float(x.n * x.h * x.w * x.c * 2 + 3 * k_real * pc.k * 2 + x.n * x.h * x.w * 3 * 16)
        self.keep.append(arr)
        self._record(self.lib.ymi_plan_add_head_decode_group(self.handle, arr, len(xs), C.byref(d)), name, kind="conv", flops=flops, bytes=nbytes,
                     ref_convs=len(xs), tile=0, shape=f"{len(xs)} levels -> {3 * k_real} k1x1 s1 +decode (one launch)")

    def post_finish(self, d: PostDesc, total_anchors: int) -> None:
        self._record(self.lib.ymi_plan_add_post_finish(self.handle, C.byref(d)), "postprocess", kind="post", flops=0.0, bytes=0.0, shape=f"A={total_anchors}")

    def stem_from_planar(self, images: Sequence[Tensor], stream: Optional[torch.cuda.Stream] = None, ptrs: Optional[bytes] = None) -> int:
        """op 0 (the stem conv in super-pixel form) computed straight from planar (3, H, W) images of the compute dtype:
        identity-size batches skip the letterbox pass and its NHWC4 round trip (ymi_conv_stem_planar).  When op 1 is
        Conv(32, 64, 3, 2, 1) over op 0's output (yolov5s: darknetv6.py:81, :85-86) both run as ONE launch and the stem's output
        never reaches memory (ymi_stem_body1_planar; YOLORT_AMD_FUSE_STEM=0 keeps them apart).  Returns the number of leading plan
        ops it has covered (1 or 2): the caller runs the plan from there."""
        d = self.conv_descs[0]
        # `ptrs`: the images' data pointers as packed 64-bit words (the C scan of the batch, YOLOv5.forward_async)
        ptrs = (C.c_void_p * len(images)).from_buffer_copy(ptrs) if ptrs is not None else (C.c_void_p * len(images))(*[im.data_ptr() for im in images])
        if self.stem_body1_fusable():
            check(self.lib.ymi_stem_body1_planar(C.byref(d), C.byref(self.conv_descs[1]), ptrs, len(images), _lib.stream_ptr(stream)), "ymi_stem_body1_planar")
            return 2
        check(self.lib.ymi_conv_stem_planar(C.byref(d), ptrs, len(images), _lib.stream_ptr(stream)), "ymi_conv_stem_planar")
        return 1

    def stem_body1_fusable(self) -> bool:
        if os.environ.get("YOLORT_AMD_FUSE_STEM", "1") == "0" or self.fp32:
            return False
        d0, d1 = self.conv_descs.get(0), self.conv_descs.get(1)
        if d0 is None or d1 is None or 1 not in self.io or self.io[1]["x"] is None or self.io[0]["y"] is None:
            return False
        return (d0.cout == 32 and d0.out_dtype == d0.dtype and d0.act == ACT_SILU and self.io[1]["x"].ptr == self.io[0]["y"].ptr and self.io[1]["x"].cs == 32
                and d1.cin == 32 and d1.cout == 64 and (d1.kh, d1.kw, d1.sh, d1.sw, d1.ph, d1.pw) == (3, 3, 2, 2, 1, 1) and d1.k_pad == 288 and d1.act == ACT_SILU
                and not d1.res and not d1.chain_w and d1.cout_split == 0 and d1.y2_mode == 0 and d1.out_dtype == d1.dtype == d0.dtype and d1.y_cstride % 8 == 0)

    def set_fuse_stem(self, on: bool) -> bool:
        """Dynamic-shape streams (ops 0 and 1 read the letterboxed canvas): run the two as ONE launch whenever a run covers both
        (ymi_plan_set_fuse_stem; op indices stay, the stem's output buffer is then not written).  Returns what is in effect."""
        on = bool(on) and self.stem_body1_fusable() and self.conv_descs[0].x_cstride == 8 and bool(self.conv_descs[0].x)
        check(self.lib.ymi_plan_set_fuse_stem(self.handle, 1 if on else 0), "ymi_plan_set_fuse_stem")
        self.fuse_stem = on
        return on

    def stem_planar_ok(self, images: Sequence[Tensor], canvas_hw: Tuple[int, int], scan=None) -> bool:
        """`scan`: the C pass over the image list (torch_ext/sig_ext.cpp images(): one device, one dtype; uniform shape, contiguity, 16-byte alignment) in place of the per-image reads"""
        d = self.conv_descs.get(0)
        if self.fp32 or d is None or not d.zeros or not (d.cin == 8 and d.kh == 6 and d.kw == 3 and d.sh == 2 and d.sw == 1 and d.cout_pad <= 64 and not d.res):
            return False
        hb, wb = canvas_hw
        if wb % 8 or d.h != hb or d.w_in * 2 != wb or len(images) != d.n:
            return False
        if scan is not None:
            return scan[3] == (3, hb, wb) and scan[5] and scan[6] and scan[1] == self.device.index and images[0].dtype == self.dtype
        return all(im.device == self.device and im.dtype == self.dtype and tuple(im.shape) == (3, hb, wb) and im.is_contiguous() and im.data_ptr() % 16 == 0
                   for im in images)

    # ---- export ----
    def export(self, path: str, io: Optional[Dict[int, Tensor]] = None) -> int:
        """Writes the recorded plan as a self-contained file (ymi_plan_export, include/yolort_amd.h): every launch descriptor with its pointers rewritten as (region,
        offset), and the contents of the constant regions -- packed weights, biases, im2col tables, the strip kernel's weight streams.  A consumer without Python loads
        it with ymi_plan_import and replays it with ymi_plan_run.  `io`: {YMI_TAG_*: tensor} names the regions the consumer talks to
        (the input canvas, the rescale rows, the detection arrays).  Returns the number of regions.  The reference's counterpart is the TorchScript / ONNX export of
        yolort/relay + yolort/runtime (out of scope: SURVEY.md 8 row f3 asks for the in-scope equivalent)."""
        from ._lib import PlanRegion, REGION_CONST, REGION_IO, REGION_SCRATCH
        tensors: List[Tuple[Tensor, int]] = []   # (tensor, kind)

        def add(t, kind):
            if isinstance(t, Tensor) and t.numel() > 0 and t.device.type == self.device.type:
                tensors.append((t, kind))

        def walk(o):
            if isinstance(o, Tensor):
                add(o, REGION_SCRATCH)
            elif isinstance(o, PackedConv):
                add(o.w, REGION_CONST)
                add(o.bias, REGION_CONST)
                for kt in o._ktabs.values():
                    add(kt, REGION_CONST)
            elif isinstance(o, PostBuffers):
                for t in (o.boxes, o.scores, o.labels, o.status_count, o.slab, o.ws, o.rescale):
                    add(t, REGION_SCRATCH)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)

        walk(self.keep)
        add(self.zeros, REGION_SCRATCH)
        const_ids = getattr(self, "_const_tensors", set())
        tags = {int(t.untyped_storage().data_ptr()): tag for tag, t in (io or {}).items()}
        seen: Dict[int, Tuple[int, int, int]] = {}   # storage base -> (bytes, kind, tag)
        for t, kind in tensors:
            st = t.untyped_storage()
            base, nbytes = int(st.data_ptr()), int(st.nbytes())
            if id(t) in const_ids:
                kind = REGION_CONST
            tag = tags.get(base, 0)
            if tag:
                kind = REGION_IO
            old = seen.get(base)
            if old is None or kind == REGION_CONST or (kind == REGION_IO and old[1] == REGION_SCRATCH):
                seen[base] = (max(nbytes, old[0]) if old else nbytes, kind, tag or (old[2] if old else 0))
        regs = (PlanRegion * len(seen))()
        for i, (base, (nbytes, kind, tag)) in enumerate(sorted(seen.items())):
            regs[i].base, regs[i].bytes, regs[i].kind, regs[i].tag = base, nbytes, kind, tag
        check(self.lib.ymi_plan_export(self.handle, regs, len(seen), path.encode(), _lib.stream_ptr()), "ymi_plan_export")
        return len(seen)

    # ---- execution ----
    @property
    def num_ops(self) -> int:
        return self.lib.ymi_plan_num_ops(self.handle)

    def run(self, first: int = 0, last: int = -1, graph: bool = False, stream: Optional[torch.cuda.Stream] = None) -> None:
        check(self.lib.ymi_plan_run(self.handle, first, last, 1 if graph else 0, _lib.stream_ptr(stream)), "ymi_plan_run")

    def profile(self, iters: int = 5) -> List[Tuple[str, float, dict]]:
        ms = (C.c_float * self.num_ops)()
        check(self.lib.ymi_plan_profile(self.handle, iters, ms, _lib.stream_ptr()), "ymi_plan_profile")
        return [(self.names[i], float(ms[i]), self.meta[i]) for i in range(self.num_ops)]


class PostBuffers:
    """Fixed-shape detection slab + workspace of one post-process op."""

    def __init__(self, plan: Plan, n: int, k: int, total_anchors: int, cand_cap: int, rescale: Optional[Tensor]):
        dev = plan.device
        self.n, self.k, self.cand_cap = n, k, cand_cap
        self.boxes = torch.zeros(n, k, 4, device=dev, dtype=torch.float32)
        self.scores = torch.zeros(n, k, device=dev, dtype=torch.float32)
        self.labels = torch.zeros(n, k, device=dev, dtype=torch.int64)
        # status words and per-image counts share one buffer: ONE device-to-host copy per batch brings both (include/yolort_amd.h ymi_post_desc.status, ABI 5: 8 words)
        self.status_count = torch.zeros(8 + n, device=dev, dtype=torch.int32)
        self.status, self.count = self.status_count[:8], self.status_count[8:]
        # the packed wire slab of yolort_amd/dist.py, written by the top-k kernel itself (ymi_post_desc.out_slab): [boxes 4K | scores K | labels K | count] per image
        self.slab = torch.zeros(n, 6 * k + 1, device=dev, dtype=torch.float32)
        self.rescale = rescale
        nbytes = plan.lib.ymi_postprocess_ws_bytes(n, total_anchors, cand_cap)
        self.ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        plan.bytes_allocated += nbytes
