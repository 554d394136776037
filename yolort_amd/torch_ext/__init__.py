"""Optional C++ operator registration (`yolort_amd::nms`, `yolort_amd::batched_nms`) for LibTorch consumers -- see yolort_amd_ops.cpp.

    python -m yolort_amd.torch_ext          builds yolort_amd/lib/libyolort_amd_torch.so in-tree (links libyolort_amd.so and libtorch)
    yolort_amd.torch_ext.load()             builds if needed and `torch.ops.load_library`s it -> torch.ops.yolort_amd.nms(...)

Nothing in the Python package needs it (the Python surface calls the C ABI through ctypes); it exists for C++ programs that used to link libtorchvision for
`torchvision::nms` (reference test/tracing/CMakeLists.txt:5,13-18).
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB = os.path.join(_PKG, "lib", "libyolort_amd_torch.so")


def build(force: bool = False) -> str:
    import torch
    from torch.utils import cpp_extension as ce

    from .. import _build

    core = _build.build(force=False, verbose=False)
    src = os.path.join(_HERE, "yolort_amd_ops.cpp")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(src), os.path.getmtime(core)):
        return LIB
    inc = [f"-I{p}" for p in ce.include_paths(device_type="cuda")] + [f"-I{os.path.join(os.path.dirname(_PKG), 'include')}", "-I/opt/rocm/include"]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cxx = shutil.which("g++") or shutil.which("c++")   # host code only (no kernels here): the system C++ compiler, HIP headers for the stream type
    if cxx is None:
        raise RuntimeError("building libyolort_amd_torch.so needs a host C++ compiler (g++ / c++ not found on PATH)")
    cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, src,
           "-o", LIB, f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip", f"-L{os.path.dirname(core)}", "-lyolort_amd",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libyolort_amd_torch.so failed:\n" + r.stderr[-3000:])
    return LIB


SIG_LIB = os.path.join(_PKG, "lib", "_ymi_sig.so")


def build_sig(force: bool = False) -> str:
    """yolort_amd/lib/_ymi_sig.so: hipmodule.weights_signature's per-batch walk as one C call (sig_ext.cpp; a CPython module linked against libtorch_python).
    Optional -- hipmodule loads it when the file is there and walks the dicts in Python otherwise."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce

    src = os.path.join(_HERE, "sig_ext.cpp")
    if not force and os.path.exists(SIG_LIB) and os.path.getmtime(SIG_LIB) >= os.path.getmtime(src):
        return SIG_LIB
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("building _ymi_sig.so needs a host C++ compiler (g++ / c++ not found on PATH)")
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    os.makedirs(os.path.dirname(SIG_LIB), exist_ok=True)
    cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-shared", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, src, "-o", SIG_LIB,
           f"-L{libdir}", "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building _ymi_sig.so failed:\n" + r.stderr[-3000:])
    return SIG_LIB


def load() -> str:
    import torch

    lib = build()
    torch.ops.load_library(lib)
    return lib


if __name__ == "__main__":
    print(build(force=True))
    print(build_sig(force=True))
