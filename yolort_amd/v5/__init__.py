"""Mirror of the reference's `yolort.v5` import surface for the hot-path blocks only."""
from .models.common import C3, SPP, SPPF, Bottleneck, Conv, autopad  # noqa: F401
