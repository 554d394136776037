"""Mirror of the reference's `yolort.v5` import surface for the hot-path blocks only."""
from .models.common import C3, SPP, SPPF, Bottleneck, BottleneckCSP, Conv, Focus, autopad, focus_transform, space_to_depth  # noqa: F401
