import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's plain-C NMS is test infrastructure; build it on demand (gcc, <1 s)
    so = os.path.join(ROOT, "oracle", "libnms_ref.so")
    if not os.path.exists(so):
        subprocess.run(["bash", os.path.join(ROOT, "oracle", "build.sh")], check=False)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
