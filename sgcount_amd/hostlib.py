"""ctypes binding of libsgcount_host.so — the C++ host side (FASTX reader, offsetter, gene map, results table,
sample names, CLI).  Return codes follow the reference's exit codes: 0 ok, 1 error (anyhow), 101 panic."""
import ctypes as C
import os

import numpy as np

from . import _ffi, build as _build


class HostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is None:
        _ffi.load()                      # libsgcount_hip.so first (shared HIP runtime, DT_NEEDED of the host lib)
        if _build._host_needs_build(_build.HOST_SO) or _build._host_needs_build(_build.CLI):
            _build.build_host()
        # SGH_HOST_LIB: another build of the host library to load instead (the AddressSanitizer build of tools/asan_host_tests.sh)
        L = C.CDLL(os.environ.get("SGH_HOST_LIB") or _build.HOST_SO)
        L.sgh_last_error.restype = C.c_char_p
        for name in ("sgh_cli", "sgh_entropy_offset_group", "sgh_positional_entropy", "sgh_minimize_mse",
                     "sgh_generate_sample_names", "sgh_genemap_get", "sgh_genemap_missing", "sgh_generate_columns",
                     "sgh_format_results", "sgh_library_info", "sgh_fastx_stats", "sgh_text_feeder_walk", "sgh_scan_records", "sgh_text_feeder_drain"):
            getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def _chk(rc):
    if rc:
        raise HostError(rc, load().sgh_last_error().decode("utf-8", "replace"))


def _blob(strings):
    return b"".join((x if isinstance(x, bytes) else x.encode()) + b"\0" for x in strings)


def cli_path():
    load()
    return _build.CLI


def entropy_offset_group(library_path, inputs, subsample=5000):
    n = len(inputs)
    rev = (C.c_int * n)()
    idx = (C.c_uint64 * n)()
    _chk(load().sgh_entropy_offset_group(library_path.encode(), _blob(inputs), n, C.c_uint64(subsample), rev, idx))
    return [(bool(rev[i]), int(idx[i])) for i in range(n)]


def positional_entropy(path, take=2 ** 64 - 1):
    out = (C.c_double * 65536)()
    n = C.c_uint64(0)
    _chk(load().sgh_positional_entropy(path.encode(), C.c_uint64(take), out, C.c_uint64(65536), C.byref(n)))
    return list(out[: n.value])


def minimize_mse(ref, cmp):
    a = (C.c_double * len(ref))(*ref)
    b = (C.c_double * len(cmp))(*cmp)
    rev, idx = C.c_int(0), C.c_uint64(0)
    _chk(load().sgh_minimize_mse(a, C.c_uint64(len(ref)), b, C.c_uint64(len(cmp)), C.byref(rev), C.byref(idx)))
    return bool(rev.value), int(idx.value)


def generate_sample_names(paths):
    out = C.create_string_buffer(1 << 16)
    _chk(load().sgh_generate_sample_names(_blob(paths), len(paths), out, C.c_uint64(len(out))))
    return out.value.decode().split("\n") if paths else []


def genemap_get(sgrna, text=None, path=None):
    out = C.create_string_buffer(1 << 12)
    found = C.c_int(0)
    _chk(load().sgh_genemap_get(path.encode() if path else None, text, sgrna, out, C.c_uint64(len(out)), C.byref(found)))
    return out.value if found.value else None


def genemap_missing(genemap_text, library_path):
    out = C.create_string_buffer(1 << 12)
    found = C.c_int(0)
    _chk(load().sgh_genemap_missing(genemap_text, library_path.encode(), out, C.c_uint64(len(out)), C.byref(found)))
    return out.value if found.value else None


def generate_columns(names, with_genemap=False):
    out = C.create_string_buffer(1 << 16)
    _chk(load().sgh_generate_columns(_blob(names), len(names), int(with_genemap), out, C.c_uint64(len(out))))
    return out.value.decode()


def format_results(library_path, counts_per_sample, names, genemap_text=None, include_zero=False):
    flat = np.ascontiguousarray(np.array(counts_per_sample, dtype=np.uint64).reshape(-1))
    cap = 1 << 24
    out = C.create_string_buffer(cap)
    _chk(load().sgh_format_results(library_path.encode(), flat.ctypes.data_as(C.POINTER(C.c_uint64)), len(counts_per_sample),
                                   _blob(names), genemap_text, int(include_zero), out, C.c_uint64(cap)))
    return out.value.decode()


def library_info(path):
    n, size = C.c_uint64(0), C.c_uint64(0)
    _chk(load().sgh_library_info(path.encode(), C.byref(n), C.byref(size)))
    return n.value, size.value


def fastx_stats(path):
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _chk(load().sgh_fastx_stats(path.encode(), C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def text_feeder_walk(path, slice_bytes=1 << 16, threads=3, pgz_chunk=0, info=None):
    """Walks a file through the text path's byte source (TextFeeder) exactly as count() does and returns
    (parts, bytes, lines, fnv1a-64 of all pushed bytes, first byte, is_gz) — is_gz is "bgzf" for a BGZF file (gzip members
    inflated in parallel).  pgz_chunk: compressed bytes per unit of work of the parallel decoder of plain gzip streams (0 = from
    the file size); info (a dict) receives {"pgz": the stream went through that decoder, "fallbacks": chunks it decoded in order}."""
    parts, nbytes, lines, fnv, fb_count = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    fb, gz = C.c_int(), C.c_int()
    _chk(load().sgh_text_feeder_walk(path.encode(), C.c_uint64(slice_bytes), C.c_uint64(threads), C.byref(parts), C.byref(nbytes),
                                     C.byref(lines), C.byref(fnv), C.byref(fb), C.byref(gz), C.c_uint64(pgz_chunk), C.byref(fb_count)))
    if info is not None:
        info.update(pgz=bool(gz.value & 4), fallbacks=fb_count.value)
    return parts.value, nbytes.value, lines.value, fnv.value, fb.value, ("bgzf" if gz.value & 2 else bool(gz.value & 1))


def text_feeder_drain(path, slice_bytes=32 << 20, threads=8, pgz_chunk=0):
    """Runs a file through the byte source of the text path and throws the bytes away: (bytes, newlines, Σ worker busy seconds,
    chunks the parallel gzip decoder decoded in order, kind) — kind: "plain" | "gz" (one thread) | "pgz" (parallel gzip) | "bgzf"."""
    nbytes, lines, fb, busy, kind = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_double(), C.c_int()
    _chk(load().sgh_text_feeder_drain(path.encode(), C.c_uint64(slice_bytes), C.c_uint64(threads), C.c_uint64(pgz_chunk), C.byref(nbytes),
                                      C.byref(lines), C.byref(busy), C.byref(fb), C.byref(kind)))
    k = kind.value
    return nbytes.value, lines.value, busy.value, fb.value, ("bgzf" if k & 2 else "pgz" if k & 4 else "gz" if k & 1 else "plain")


def scan_records(path, L, reverse=False, offset=0, recursion=True, threads=3, block_bytes=1 << 16, cap=None, source=0):
    """All packed records of a plain FASTQ file as the scan path produces them (FastqScanner: memory-mapped text, several
    threads, blocks in order).  Returns (records as a uint64 array [n, words], total lines) or None if the scanner declines
    the file (not plain FASTQ text).  Raises HostError 101 on a malformed / truncated record.  source: 0 auto, 1 the memory
    mapping, 2 pread() into the threads' buffers."""
    words = 2 if L > 23 else 1
    if cap is None:
        cap = os.path.getsize(path) // 4 + 16          # a record takes at least 4 newlines... of text
    out = np.zeros((cap, words), dtype=np.uint64)
    n, lines, usable = C.c_uint64(), C.c_uint64(), C.c_int()
    _chk(load().sgh_scan_records(path.encode(), C.c_uint32(L), int(bool(reverse)), C.c_uint32(offset), int(bool(recursion)),
                                 C.c_uint64(threads), C.c_uint64(block_bytes), out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 C.c_uint64(cap), C.byref(n), C.byref(lines), C.byref(usable), int(source)))
    if not usable.value:
        return None
    return out[: n.value], lines.value


def cli(argv):
    """Runs the sgcount-hip command line in-process (sgh::cli_main); returns its exit code (0 ok, 1 error,
    101 panic, 2 usage)."""
    args = [b"sgcount-hip"] + [a.encode() if isinstance(a, str) else a for a in argv]
    arr = (C.c_char_p * len(args))(*args)
    return load().sgh_cli(len(args), arr)


def count(library_path, input_paths, sample_names=None, output_path=None, offset=None, exact=False, genemap=None,
          position_recursion=True, include_zero=False, quiet=True, reverse=False, threads=1, subsample=None):
    """count() of the reference (src/count.rs:74-148) with main()'s option handling (src/main.rs:142-203):
    offset = None → entropy auto-offset; an int → Forward(offset) (Reverse if reverse=True)."""
    argv = ["-l", library_path, "-i", *input_paths, "-t", str(threads)]
    if sample_names:
        argv += ["-n", *sample_names]
    if output_path:
        argv += ["-o", output_path]
    if offset is not None:
        argv += ["-a", str(int(offset))]
    if reverse:
        argv.append("-r")
    if exact:
        argv.append("-x")
    if genemap:
        argv += ["-g", genemap]
    if not position_recursion:
        argv.append("-p")
    if include_zero:
        argv.append("-z")
    if quiet:
        argv.append("-q")
    if subsample is not None:
        argv += ["-s", str(int(subsample))]
    rc = cli(argv)
    if rc:
        raise HostError(rc, "sgcount-hip exited with code %d" % rc)
