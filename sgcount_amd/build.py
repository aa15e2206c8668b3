"""Builds libsgcount_hip.so (C ABI + gfx950 kernels) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SO = os.path.join(PKG, "libsgcount_hip.so")
SOURCES = ["sgc_api.cpp", "sgc_tables.cpp", "sgc_kernels.hip"]
HEADERS = ["sgc_format.h", "sgc_kernels.h", "sgc_tables.h", os.path.join("..", "..", "include", "sgcount_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libsgcount_hip.so (no CPU fallback exists)")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-result", "-Wno-unused-value", "-o", SO] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
