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
# experiment builds (tools/tune.py --lib ablate|stamps), never built by build(): timing-only ablation branches / phase stamps compiled in
ABLATE_SO = os.path.join(PKG, "libsgcount_hip_ablate.so")
STAMPS_SO = os.path.join(PKG, "libsgcount_hip_stamps.so")
EXPERIMENT_SOS = {"ablate": ABLATE_SO, "stamps": STAMPS_SO}
TARGETS[ABLATE_SO] = TARGETS[SO]
TARGETS[STAMPS_SO] = TARGETS[SO]
EXTRA_FLAGS = {CHECK_SO: ["-DSGC_CHECK=1"], ABLATE_SO: ["-DSGC_ABLATE=1"], STAMPS_SO: ["-DSGC_STAMPS=1"]}


def needs_build(so=SO):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(d) > t for d in _deps(so) if os.path.exists(d))


def build_one(so, force=False, verbose=False):
    """One object per source, compiled side by side (the kernels of one file take most of a minute), then linked.  The objects are
    kept under build/<library name>/ and reused while the source, every header and the flags are unchanged."""
    if not force and not needs_build(so):
        return so
    from concurrent.futures import ThreadPoolExecutor
    srcs, hdrs = TARGETS[so]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-value"] + \
            EXTRA_FLAGS.get(so, []) + os.environ.get("SGC_HIPCC_FLAGS", "").split()      # SGC_HIPCC_FLAGS: e.g. -DSGC_STAMPS=1
    odir = os.path.join(PKG, "build", os.path.basename(so))
    os.makedirs(odir, exist_ok=True)
    stamp = os.path.join(odir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in hdrs if os.path.exists(os.path.join(CSRC, h)))
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))

    def one(f):
        src, obj = os.path.join(CSRC, f), os.path.join(odir, f.replace(os.sep, "_") + ".o")
        if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            return obj
        cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, srcs))
    with open(stamp, "w") as fh:
        fh.write(" ".join(flags))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs
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
        if so not in EXPERIMENT_SOS.values():
            build_one(so, force, verbose)
    build_host(force, verbose)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
