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
CHECK_SO = os.path.join(PKG, "libsgcount_hip_check.so")     # the same sources with -DSGC_CHECK=1 (sgc_kernels.h): bounds-checked scratch indexing
HOST_SO = os.path.join(PKG, "libsgcount_host.so")
CLI = os.path.join(PKG, "bin", "sgcount-hip")
HOST_SRCS = [os.path.join("host", f) for f in ("sgh.cpp", "sgh_scan.cpp", "sgh_inflate.cpp", "sgh_cli.cpp", "sgh_capi.cpp")]
HOST_HDRS = [os.path.join("host", "sgh.hpp"), "sgc_format.h", os.path.join(INC, "sgcount_hip.h")]

TARGETS = {
    SO: (["sgc_api.cpp", "sgc_tables.cpp", "sgc_kernels.hip", "sgc_part.hip", "sgc_core.hip", "sgc_build.hip", "sgc_fastq.hip", "sgc_bytes.hip"],
         ["sgc_format.h", "sgc_device.h", "sgc_kernels.h", "sgc_tables.h", "sgc_bytes.h", "sgc_runs.h", os.path.join(INC, "sgcount_hip.h")]),
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


TARGETS[CHECK_SO] = TARGETS[SO]
EXTRA_FLAGS = {CHECK_SO: ["-DSGC_CHECK=1"]}


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
           "-Wno-unused-result", "-Wno-unused-value"] + EXTRA_FLAGS.get(so, []) + os.environ.get("SGC_HIPCC_FLAGS", "").split() + \
          ["-o", so] + [os.path.join(CSRC, f) for f in srcs]      # SGC_HIPCC_FLAGS: e.g. -DSGC_STAMPS=1 (tools/evidence.sh)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return so


def _host_needs_build(out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in HOST_SRCS + HOST_HDRS] + [SO, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_host(force=False, verbose=False):
    """C++ host side (FASTX reader, offsetter, gene map, results, CLI) — plain g++ over the C ABI."""
    cxx = shutil.which("g++") or "g++"
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-pthread"]
    link = ["-L" + PKG, "-lsgcount_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + PKG]
    srcs = [os.path.join(CSRC, f) for f in HOST_SRCS]
    if force or _host_needs_build(HOST_SO):
        cmd = [cxx] + common + ["-shared", "-DSGH_NO_MAIN", "-o", HOST_SO] + srcs + link
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    if force or _host_needs_build(CLI):
        os.makedirs(os.path.dirname(CLI), exist_ok=True)
        cmd = [cxx] + common + ["-o", CLI] + srcs[:4] + ["-L" + PKG, "-lsgcount_hip", "-lz", "-Wl,-rpath,$ORIGIN/..",
                                                         "-Wl,-rpath," + PKG]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_SO


def build(force=False, verbose=False):
    for so in TARGETS:
        build_one(so, force, verbose)
    build_host(force, verbose)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
