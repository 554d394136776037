"""ctypes binding of libyolort_amd.so (C ABI in include/yolort_amd.h).

There is NO fallback: if the shared library is missing or no MI355X is visible, every compute
entry point raises.  `import torch` happens first on purpose: the torch wheel bundles its own
libamdhip64.so (same SONAME as /opt/rocm's), and loading ours afterwards makes the dynamic loader
bind our HIP symbols to the runtime torch already initialised, so streams, events and device
pointers are shared (SURVEY.md section 7 "Two ROCm runtimes in one process").
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YOLORT_AMD_LIB") or os.path.join(_HERE, "lib", "libyolort_amd.so")   # override: tuning builds only

YMI_F16, YMI_BF16, YMI_F32, YMI_U8 = 0, 1, 2, 3
YMI_U8_HWC = 4   # ymi_letterbox input only: interleaved (h, w, 3) uint8
ACT_NONE, ACT_SILU, ACT_HARDSWISH, ACT_LEAKY = 0, 1, 2, 3   # include/yolort_amd.h YMI_ACT_*: the last two belong to the legacy r3.1 blocks
MAX_LEVELS = 4


class YmiError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("ktab", C.c_void_p), ("y", C.c_void_p), ("res", C.c_void_p),
        ("n", C.c_int32), ("h", C.c_int32), ("w_in", C.c_int32), ("cin", C.c_int32), ("x_cstride", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32), ("cout", C.c_int32), ("cout_pad", C.c_int32), ("y_cstride", C.c_int32), ("res_cstride", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32), ("k_pad", C.c_int32),
        ("act", C.c_int32), ("dtype", C.c_int32), ("out_dtype", C.c_int32), ("tile", C.c_int32),
        ("y2", C.c_void_p), ("y2_cstride", C.c_int32), ("cout_split", C.c_int32),
        ("y2_mode", C.c_int32), ("reserved0", C.c_int32),
        ("chain_w", C.c_void_p), ("chain_bias", C.c_void_p), ("chain_y", C.c_void_p), ("chain_cout", C.c_int32), ("chain_y_cstride", C.c_int32),
        ("chain_x2", C.c_void_p), ("chain_x2_cstride", C.c_int32), ("chain_k2", C.c_int32),
        ("zeros", C.c_void_p),
    ]


class C3Desc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("w12", C.c_void_p), ("b12", C.c_void_p), ("wm1", C.c_void_p), ("bm1", C.c_void_p),
        ("wm2", C.c_void_p), ("bm2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("x_cstride", C.c_int32), ("y_cstride", C.c_int32), ("dtype", C.c_int32),
        ("c_in", C.c_int32), ("c_hidden", C.c_int32), ("c_out", C.c_int32), ("n_bottlenecks", C.c_int32), ("shortcut", C.c_int32),
        ("k12_pad", C.c_int32), ("km1_pad", C.c_int32), ("km2_pad", C.c_int32), ("k3_pad", C.c_int32), ("mode", C.c_int32),
        # ABI 6: the strip kernel (csrc/c3_tile.hip)
        ("wblob", C.c_void_p), ("y1_in", C.c_void_p), ("y1_out", C.c_void_p), ("y2", C.c_void_p),
        ("y1_in_cstride", C.c_int32), ("y1_out_cstride", C.c_int32), ("y2_cstride", C.c_int32), ("reserved1", C.c_int32),
    ]


class PlanRegion(C.Structure):
    _fields_ = [("base", C.c_void_p), ("bytes", C.c_int64), ("kind", C.c_int32), ("tag", C.c_int32)]


REGION_CONST, REGION_SCRATCH, REGION_IO = 1, 2, 3
TAG_NONE, TAG_INPUT, TAG_RESCALE, TAG_BOXES, TAG_SCORES, TAG_LABELS, TAG_STATUS_COUNT, TAG_SLAB = range(8)
ABI_VERSION = 6   # include/yolort_amd.h YMI_ABI_VERSION
POST_EXACT_FULL = 1


class PostDesc(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p * MAX_LEVELS),
        ("lh", C.c_int32 * MAX_LEVELS), ("lw", C.c_int32 * MAX_LEVELS), ("lcstride", C.c_int32 * MAX_LEVELS),
        ("stride", C.c_float * MAX_LEVELS),
        ("anchors", (C.c_float * 6) * MAX_LEVELS),
        ("num_levels", C.c_int32), ("n", C.c_int32), ("num_classes", C.c_int32),
        ("score_thresh", C.c_float), ("nms_thresh", C.c_float),
        ("detections_per_img", C.c_int32),
        ("rescale", C.c_void_p),
        ("out_boxes", C.c_void_p), ("out_scores", C.c_void_p), ("out_labels", C.c_void_p), ("out_count", C.c_void_p),
        ("status", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("cand_cap", C.c_int32),
        ("flags", C.c_int32),
        ("out_slab", C.c_void_p),   # ABI 5: packed wire slab (n, 6K + 1) fp32 or NULL
    ]


_SIGS = {
    "ymi_abi_version": (C.c_int, []),
    "ymi_last_error": (C.c_char_p, []),
    "ymi_device_count": (C.c_int, []),
    "ymi_letterbox": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "ymi_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ymi_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ymi_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "ymi_c3_fused": (C.c_int, [C.POINTER(C3Desc), C.c_void_p]),
    "ymi_plan_add_c3_fused": (C.c_int, [C.c_void_p, C.POINTER(C3Desc)]),
    "ymi_c3_blob_bytes": (C.c_int64, [C.POINTER(C3Desc)]),
    "ymi_c3_pack": (C.c_int, [C.POINTER(C3Desc), C.c_void_p, C.c_void_p]),
    "ymi_c3_tile_supported": (C.c_int, [C.POINTER(C3Desc)]),
    "ymi_c3_tile_geometry": (C.c_int, [C.POINTER(C3Desc), C.POINTER(C.c_int)]),
    "ymi_conv_stem_planar": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "ymi_stem_body1_planar": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "ymi_stem_body1": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_void_p]),
    "ymi_clock_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ymi_conv_build_ktab": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "ymi_conv_f32_pick_tile": (C.c_int, [C.c_int, C.c_int]),
    "ymi_spp_pool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ymi_upsample2x": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ymi_copy_view": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ymi_act": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ymi_plan_add_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "ymi_postprocess_ws_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "ymi_postprocess": (C.c_int, [C.POINTER(PostDesc), C.c_void_p]),
    "ymi_post_begin": (C.c_int, [C.POINTER(PostDesc), C.c_void_p]),
    "ymi_post_finish": (C.c_int, [C.POINTER(PostDesc), C.c_void_p]),
    "ymi_conv_head_decode_group": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.POINTER(PostDesc), C.c_void_p]),
    "ymi_conv_head_decode": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(PostDesc), C.c_int, C.c_void_p]),
    "ymi_nms_ws_bytes": (C.c_int64, [C.c_int]),
    "ymi_batched_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ymi_plan_create": (C.c_void_p, []),
    "ymi_plan_destroy": (None, [C.c_void_p]),
    "ymi_plan_add_conv": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc)]),
    "ymi_plan_add_spp_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ymi_plan_add_upsample2x": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "ymi_plan_add_copy_view": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "ymi_plan_add_postprocess": (C.c_int, [C.c_void_p, C.POINTER(PostDesc)]),
    "ymi_plan_add_post_begin": (C.c_int, [C.c_void_p, C.POINTER(PostDesc)]),
    "ymi_plan_add_head_decode": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc), C.POINTER(PostDesc), C.c_int]),
    "ymi_plan_add_head_decode_group": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc), C.c_int, C.POINTER(PostDesc)]),
    "ymi_plan_add_post_finish": (C.c_int, [C.c_void_p, C.POINTER(PostDesc)]),
    "ymi_plan_num_ops": (C.c_int, [C.c_void_p]),
    "ymi_plan_set_fuse_stem": (C.c_int, [C.c_void_p, C.c_int]),
    "ymi_plan_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ymi_plan_profile": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "ymi_plan_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ymi_plan_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "ymi_plan_done_query": (C.c_int, [C.c_void_p]),
    "ymi_plan_done_sync": (C.c_int, [C.c_void_p]),
    "ymi_plan_export": (C.c_int, [C.c_void_p, C.POINTER(PlanRegion), C.c_int, C.c_char_p, C.c_void_p]),
    "ymi_plan_import": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(PlanRegion), C.c_int, C.POINTER(C.c_int)]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


def load(require_gpu: bool = False) -> C.CDLL:
    """Loads the shared library (once).  Raises YmiError loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YmiError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -m yolort_amd._build` "
                "(needs hipcc, cross-compiles for gfx950 without a GPU). There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.ymi_abi_version() != ABI_VERSION:
            raise YmiError(f"ABI version mismatch: library reports {lib.ymi_abi_version()}, binding expects {ABI_VERSION}")
        _lib = lib
    if require_gpu:
        if not torch.cuda.is_available():
            raise YmiError("no MI355X visible (torch.cuda.is_available() is False): yolort_amd has no CPU fallback")
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc < 0:
        msg = _lib.ymi_last_error().decode("utf-8", "replace") if _lib is not None else ""
        raise YmiError(f"{what or 'libyolort_amd'} failed (code {rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    try:
        return {torch.float16: YMI_F16, torch.bfloat16: YMI_BF16, torch.float32: YMI_F32, torch.uint8: YMI_U8}[dt]
    except KeyError:
        raise YmiError(f"unsupported dtype {dt}") from None


def stream_ptr(stream: Optional["torch.cuda.Stream"] = None) -> C.c_void_p:
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)
