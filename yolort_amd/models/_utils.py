"""Small host-side helpers shared by the model graph pieces."""
from typing import Optional


def _make_divisible(v: float, divisor: int, min_value: Optional[int] = None) -> int:
    """Channel rounding rule of the reference (yolort/models/_utils.py:10-23): nearest multiple of
    `divisor`, never below `min_value`, never more than 10% below `v`."""
    floor = divisor if min_value is None else min_value
    rounded = int(v + divisor / 2) // divisor * divisor
    out = rounded if rounded > floor else floor
    return out + divisor if out < 0.9 * v else out
