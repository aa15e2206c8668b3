"""Builds the in-tree native libraries with hipcc for gfx950:

  libsgcount_hip.so    C ABI + count/pack kernels   (include/sgcount_hip.h)
  libsgcount_synth.so  synthetic workload generator (include/sgcount_synth.h; bench/tests only)

hipcc cross-compiles for gfx950 without a GPU; the built .so files travel to the GPU box with the
repo snapshot (they are git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(os.path.dirname(PKG), "include")
SO = os.path.join(PKG, "libsgcount_hip.so")
SYNTH_SO = os.path.join(PKG, "libsgcount_synth.so")

TARGETS = {
    SO: (["sgc_api.cpp", "sgc_tables.cpp", "sgc_kernels.hip", "sgc_part.hip"],
         ["sgc_format.h", "sgc_device.h", "sgc_kernels.h", "sgc_tables.h", os.path.join(INC, "sgcount_hip.h")]),
    SYNTH_SO: (["sgc_synth.hip"], ["sgc_format.h", "sgc_synth.h", os.path.join(INC, "sgcount_synth.h")]),
}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the HIP libraries (no CPU fallback exists)")


def _deps(so):
    srcs, hdrs = TARGETS[so]
    return [os.path.join(CSRC, f) for f in srcs + hdrs] + [os.path.abspath(__file__)]


def needs_build(so=SO):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(d) > t for d in _deps(so) if os.path.exists(d))


def build_one(so, force=False, verbose=False):
    if not force and not needs_build(so):
        return so
    srcs, _ = TARGETS[so]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-result", "-Wno-unused-value", "-o", so] + [os.path.join(CSRC, f) for f in srcs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return so


def build(force=False, verbose=False):
    for so in TARGETS:
        build_one(so, force, verbose)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
