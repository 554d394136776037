"""Builds yolort_amd/lib/libyolort_amd.so with hipcc for gfx950 (in-tree, no torch extension
machinery, no hipify).  Usage: python -m yolort_amd._build [--force]"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.environ.get("YOLORT_AMD_BUILD_OUT") or os.path.join(LIBDIR, "libyolort_amd.so")   # override: tuning builds only
SOURCES = ["api.cpp", "conv_igemm.hip", "conv3x3_halo.hip", "conv_halo8.hip", "conv_igemm8.hip", "conv1x1_stream.hip", "conv3x3_c32.hip", "conv3x3_res.hip", "conv3x3_rw.hip", "conv3x3_rw2.hip", "conv3x3_rs.hip", "c3_fused32.hip", "c3_tile.hip", "stem_body1_fused.hip", "conv_stem.hip", "conv_f32.hip", "conv_f32_pipe.hip", "preproc_pool.hip", "postprocess.hip"]
# the conv tile x dtype space and the fused heads are instantiated in their own translation units (parallel build)
INST_SOURCES = sorted(f for f in os.listdir(CSRC) if (f.startswith("conv_inst_") or f.startswith("head_inst_")) and f.endswith(".hip"))
MONOLITHIC = "-DYMI_STAMPS" in os.environ.get("YOLORT_AMD_BUILD_FLAGS", "")   # the timeline instrumentation keeps one device symbol: single TU
SOURCES = SOURCES + ([] if MONOLITHIC else INST_SOURCES)
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join(os.path.dirname(PKG), "include", "yolort_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]
FLAGS += os.environ.get("YOLORT_AMD_BUILD_FLAGS", "").split()
if MONOLITHIC:
    FLAGS.append("-DYMI_MONOLITHIC")   # e.g. -DYMI_STAMPS for tools/stamp_conv.py


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _digest() -> str:
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = LIB + ".stamp" if os.environ.get("YOLORT_AMD_BUILD_OUT") else os.path.join(LIBDIR, "build.stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []

    hdr = hashlib.sha256()
    for f in HEADERS:
        hdr.update(open(f, "rb").read())
    hdr.update(" ".join(FLAGS).encode())
    hdr_digest = hdr.hexdigest()

    def compile_one(src: str) -> str:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + (".dbg.o" if os.environ.get("YOLORT_AMD_BUILD_OUT") else ".o"))
        # incremental: an object is reused when its own source, every header and the flags are unchanged
        odig = hashlib.sha256(open(os.path.join(CSRC, src), "rb").read() + hdr_digest.encode()).hexdigest()
        ostamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            return obj
        cmd = [hipcc, *FLAGS, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr[-2000:], file=sys.stderr)
        open(ostamp, "w").write(odig)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) // 1024} KB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
