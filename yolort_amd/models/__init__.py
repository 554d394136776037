"""Public model API, same names and kwargs as `yolort.models` (reference models/__init__.py:24-185)."""
from typing import Any

from .yolo import YOLO
from .yolov5 import YOLOv5

__all__ = ["YOLO", "YOLOv5", "yolov5n", "yolov5n6", "yolov5s", "yolov5s6", "yolov5m", "yolov5m6", "yolov5l", "yolov5ts"]


def _make(arch_r60: str, upstream_version: str, export_friendly: bool, only_r60: bool, **kwargs: Any) -> YOLOv5:
    if upstream_version == "r6.0":
        model = YOLOv5(arch=arch_r60, **kwargs)
    elif upstream_version in ("r3.1", "r4.0") and not only_r60:   # reference models/__init__.py:51-54 (yolov5s / m / l)
        model = YOLOv5(arch=arch_r60.replace("_r60", "_r31" if upstream_version == "r3.1" else "_r40"), **kwargs)
    elif only_r60:
        raise NotImplementedError("Currently only supports r6.0 version")
    else:
        raise NotImplementedError("Currently doesn't support this versions.")
    if export_friendly:
        raise NotImplementedError("export_friendly targets the ONNX/TVM exporters, which this MI355X-native build drops")
    return model


def yolov5n(upstream_version: str = "r6.0", export_friendly: bool = False, **kwargs: Any):
    return _make("yolov5_darknet_pan_n_r60", upstream_version, export_friendly, True, **kwargs)


def yolov5s(upstream_version: str = "r6.0", export_friendly: bool = False, **kwargs: Any):
    return _make("yolov5_darknet_pan_s_r60", upstream_version, export_friendly, False, **kwargs)


def yolov5m(upstream_version: str = "r6.0", export_friendly: bool = False, **kwargs: Any):
    return _make("yolov5_darknet_pan_m_r60", upstream_version, export_friendly, False, **kwargs)


def yolov5l(upstream_version: str = "r6.0", export_friendly: bool = False, **kwargs: Any):
    return _make("yolov5_darknet_pan_l_r60", upstream_version, export_friendly, False, **kwargs)


def yolov5n6(upstream_version: str = "r6.0", export_friendly: bool = False, **kwargs: Any):
    kwargs.setdefault("size_divisible", 64)
    return _make("yolov5_darknet_pan_n6_r60", upstream_version, export_friendly, True, **kwargs)


def yolov5s6(upstream_version: str = "r6.0", export_friendly: bool = False, **kwargs: Any):
    kwargs.setdefault("size_divisible", 64)
    return _make("yolov5_darknet_pan_s6_r60", upstream_version, export_friendly, True, **kwargs)


def yolov5m6(upstream_version: str = "r6.0", export_friendly: bool = False, **kwargs: Any):
    kwargs.setdefault("size_divisible", 64)
    return _make("yolov5_darknet_pan_m6_r60", upstream_version, export_friendly, True, **kwargs)


def yolov5ts(upstream_version: str = "r4.0", export_friendly: bool = False, **kwargs: Any):
    raise NotImplementedError("yolov5ts (transformer neck, r4.0 only) is out of the MI355X hot-path scope")
