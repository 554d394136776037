"""TEST INFRASTRUCTURE -- imports the UNMODIFIED reference read-only from /root/reference.

Only usable in the build container (the GPU box has no /root/reference); used by
tests/golden/make_golden.py and oracle/make_synth_bn.py, never at GPU-test/bench run time.
"""
import os
import sys

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "yolort"))


def load_reference():
    """Returns the reference's `yolort` package (torchvision stand-in installed first)."""
    if not reference_available():
        raise RuntimeError("reference not present at /root/reference")
    from oracle import _tv_compat

    _tv_compat.install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    import yolort  # noqa: F401
    import yolort.models  # noqa: F401
    import yolort.models.yolo  # noqa: F401

    return yolort
