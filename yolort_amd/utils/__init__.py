"""Host-side helpers (reference yolort/utils/__init__.py:48-60 `contains_any_tensor`)."""
from typing import Any, Type

from torch import Tensor


def contains_any_tensor(value: Any, dtype: Type = Tensor) -> bool:
    """True when `value` is, or (recursively) holds, an instance of `dtype`."""
    if isinstance(value, dtype):
        return True
    if isinstance(value, (list, tuple)):
        return any(contains_any_tensor(v, dtype=dtype) for v in value)
    if isinstance(value, dict):
        return any(contains_any_tensor(v, dtype=dtype) for v in value.values())
    return False
