"""ctypes binding of the CPU oracle (oracle/libsgcount_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by sgcount_amd/.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "libsgcount_oracle.so")

POS_PLUS, POS_MINUS, POS_CENTERED, POS_NULL = 0, 1, 2, 3
E_DUPLICATE_SEQ, E_INCONSISTENT, E_EMPTY, E_FORMAT, E_ARG, E_SHORT, E_NAN = -1, -2, -3, -4, -5, -6, -7


def build_oracle():
    src = os.path.join(ORACLE_DIR, "sgcount_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        L = _lib
        u8p, sz, vp = C.c_char_p, C.c_size_t, C.c_void_p
        L.orc_library_from_text.restype = vp
        L.orc_library_from_text.argtypes = [u8p, sz, C.POINTER(C.c_int)]
        L.orc_library_free.argtypes = [vp]
        L.orc_library_size.restype = sz
        L.orc_library_size.argtypes = [vp]
        L.orc_library_n.restype = sz
        L.orc_library_n.argtypes = [vp]
        L.orc_library_contains.restype = vp
        L.orc_library_contains.argtypes = [vp, u8p, sz, C.POINTER(sz)]
        L.orc_library_seq.restype = vp
        L.orc_library_seq.argtypes = [vp, sz]
        L.orc_library_id.restype = vp
        L.orc_library_id.argtypes = [vp, sz, C.POINTER(sz)]
        L.orc_permuter_new.restype = vp
        L.orc_permuter_new.argtypes = [vp]
        L.orc_permuter_from_seqs.restype = vp
        L.orc_permuter_from_seqs.argtypes = [u8p, sz, sz]
        L.orc_permuter_free.argtypes = [vp]
        L.orc_permuter_contains.restype = vp
        L.orc_permuter_contains.argtypes = [vp, u8p, sz]
        L.orc_permuter_map_len.restype = sz
        L.orc_permuter_map_len.argtypes = [vp]
        L.orc_permuter_null_len.restype = sz
        L.orc_permuter_null_len.argtypes = [vp]
        L.orc_permuter_null_contains.restype = C.c_int
        L.orc_permuter_null_contains.argtypes = [vp, u8p, sz]
        L.orc_bounds.restype = C.c_int
        L.orc_bounds.argtypes = [sz, sz, sz, C.c_int, C.POINTER(sz), C.POINTER(sz)]
        L.orc_counter_new.restype = vp
        L.orc_counter_new.argtypes = [vp, vp, C.c_int, sz, sz, C.c_int]
        L.orc_counter_feed_text.restype = C.c_int
        L.orc_counter_feed_text.argtypes = [vp, u8p, sz]
        L.orc_counter_feed_seq.argtypes = [vp, u8p, sz]
        L.orc_counter_free.argtypes = [vp]
        L.orc_counter_get_value.restype = C.c_uint64
        L.orc_counter_get_value.argtypes = [vp, u8p, sz]
        L.orc_counter_total_reads.restype = C.c_uint64
        L.orc_counter_total_reads.argtypes = [vp]
        L.orc_counter_matched_reads.restype = C.c_uint64
        L.orc_counter_matched_reads.argtypes = [vp]
        L.orc_counter_fraction_mapped.restype = C.c_double
        L.orc_counter_fraction_mapped.argtypes = [vp]
        L.orc_counter_table.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
        L.orc_count_text.restype = C.c_int
        L.orc_count_text.argtypes = [u8p, sz, u8p, sz, C.c_int, sz, C.c_int, C.c_int,
                                     C.POINTER(C.c_uint64), sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_generate_sample_names.restype = C.c_int
        L.orc_generate_sample_names.argtypes = [u8p, sz, C.c_char_p, sz, C.POINTER(C.c_int)]
        L.orc_genemap_from_text.restype = vp
        L.orc_genemap_from_text.argtypes = [u8p, sz, C.POINTER(C.c_int)]
        L.orc_genemap_free.argtypes = [vp]
        L.orc_genemap_get.restype = vp
        L.orc_genemap_get.argtypes = [vp, u8p, sz, C.POINTER(sz)]
        L.orc_genemap_missing.restype = C.c_long
        L.orc_genemap_missing.argtypes = [vp, vp]
        L.orc_generate_columns.restype = C.c_int
        L.orc_generate_columns.argtypes = [u8p, sz, C.c_int, C.c_char_p, sz]
        L.orc_format_results.restype = C.c_long
        L.orc_format_results.argtypes = [vp, C.POINTER(C.c_uint64), sz, u8p, vp, C.c_int, C.c_char_p, sz]
        L.orc_position_counts.restype = C.c_int
        L.orc_position_counts.argtypes = [u8p, sz, sz, C.POINTER(C.c_double), sz, C.POINTER(sz)]
        L.orc_positional_entropy.restype = C.c_int
        L.orc_positional_entropy.argtypes = [u8p, sz, sz, C.POINTER(C.c_double), sz, C.POINTER(sz)]
        L.orc_minimize_mse.restype = C.c_int
        L.orc_minimize_mse.argtypes = [C.POINTER(C.c_double), sz, C.POINTER(C.c_double), sz,
                                       C.POINTER(C.c_int), C.POINTER(sz)]
        L.orc_entropy_offset.restype = C.c_int
        L.orc_entropy_offset.argtypes = [u8p, sz, u8p, sz, sz, C.POINTER(C.c_int), C.POINTER(sz)]
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


class Library:
    """library.rs Library built from FASTA/FASTQ text."""

    def __init__(self, text: bytes):
        err = C.c_int(0)
        self.h = lib().orc_library_from_text(text, len(text), C.byref(err))
        if not self.h:
            raise OracleError(err.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_library_free(self.h)
            self.h = None

    def size(self):
        return lib().orc_library_size(self.h)

    def n(self):
        return lib().orc_library_n(self.h)

    def contains(self, tok: bytes):
        n = C.c_size_t(0)
        p = lib().orc_library_contains(self.h, tok, len(tok), C.byref(n))
        return C.string_at(p, n.value) if p else None

    def seqs(self):
        L = self.size()
        return [C.string_at(lib().orc_library_seq(self.h, i), L) for i in range(self.n())]

    def ids(self):
        out = []
        for i in range(self.n()):
            n = C.c_size_t(0)
            p = lib().orc_library_id(self.h, i, C.byref(n))
            out.append(C.string_at(p, n.value))
        return out


class Permuter:
    """permutes.rs Permuter."""

    def __init__(self, library: Library = None, seqs=None):
        if library is not None:
            self.L = library.size()
            self.h = lib().orc_permuter_new(library.h)
        else:
            self.L = len(seqs[0])
            flat = b"".join(seqs)
            self.h = lib().orc_permuter_from_seqs(flat, len(seqs), self.L)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_permuter_free(self.h)
            self.h = None

    def contains(self, tok: bytes):
        p = lib().orc_permuter_contains(self.h, tok, len(tok))
        return C.string_at(p, self.L) if p else None

    def map_len(self):
        return lib().orc_permuter_map_len(self.h)

    def null_len(self):
        return lib().orc_permuter_null_len(self.h)

    def null_contains(self, tok: bytes):
        return bool(lib().orc_permuter_null_contains(self.h, tok, len(tok)))


def bounds(seq_len, offset, size, position):
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    ok = lib().orc_bounds(seq_len, offset, size, position, C.byref(lo), C.byref(hi))
    return (lo.value, hi.value) if ok else None


class Counter:
    """counter.rs Counter (new → feed → accessors)."""

    def __init__(self, library: Library, permuter, reverse: bool, offset: int, size: int, position_recursion: bool):
        self.library, self.permuter = library, permuter  # keep alive
        self.h = lib().orc_counter_new(library.h, permuter.h if permuter else None, int(reverse), offset, size,
                                       int(position_recursion))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_counter_free(self.h)
            self.h = None

    def feed_text(self, text: bytes):
        rc = lib().orc_counter_feed_text(self.h, text, len(text))
        if rc:
            raise OracleError(rc)
        return self

    def feed_seq(self, seq: bytes):
        lib().orc_counter_feed_seq(self.h, seq, len(seq))
        return self

    def get_value(self, ident: bytes):
        return lib().orc_counter_get_value(self.h, ident, len(ident))

    def total_reads(self):
        return lib().orc_counter_total_reads(self.h)

    def matched_reads(self):
        return lib().orc_counter_matched_reads(self.h)

    def fraction_mapped(self):
        return lib().orc_counter_fraction_mapped(self.h)

    def table(self):
        n = self.library.n()
        out = (C.c_uint64 * n)()
        lib().orc_counter_table(self.h, self.library.h, out)
        return list(out)


def count_text(lib_text: bytes, reads_text: bytes, reverse=False, offset=0, exact=False, position_recursion=True):
    """count.rs count() for one sample → (counts in library order, total, matched)."""
    n = Library(lib_text).n()
    out = (C.c_uint64 * n)()
    tot, mat = C.c_uint64(0), C.c_uint64(0)
    rc = lib().orc_count_text(lib_text, len(lib_text), reads_text, len(reads_text), int(reverse), offset, int(exact),
                              int(position_recursion), out, n, C.byref(tot), C.byref(mat))
    if rc:
        raise OracleError(rc)
    return list(out), tot.value, mat.value


def position_counts(text: bytes, take=None):
    cap = 1 << 12
    out = (C.c_double * (cap * 4))()
    n = C.c_size_t(0)
    rc = lib().orc_position_counts(text, len(text), (2 ** 64 - 1) if take is None else take, out, cap, C.byref(n))
    if rc:
        raise OracleError(rc)
    return [list(out[i * 4: i * 4 + 4]) for i in range(n.value)]


def positional_entropy(text: bytes, take=None):
    cap = 1 << 16
    out = (C.c_double * cap)()
    n = C.c_size_t(0)
    rc = lib().orc_positional_entropy(text, len(text), (2 ** 64 - 1) if take is None else take, out, cap, C.byref(n))
    if rc:
        raise OracleError(rc)
    return list(out[: n.value])


def minimize_mse(ref, cmp):
    a = (C.c_double * len(ref))(*ref)
    b = (C.c_double * len(cmp))(*cmp)
    rev, idx = C.c_int(0), C.c_size_t(0)
    rc = lib().orc_minimize_mse(a, len(ref), b, len(cmp), C.byref(rev), C.byref(idx))
    if rc:
        raise OracleError(rc)
    return bool(rev.value), idx.value


def entropy_offset(lib_text: bytes, reads_text: bytes, subsample=5000):
    rev, idx = C.c_int(0), C.c_size_t(0)
    rc = lib().orc_entropy_offset(lib_text, len(lib_text), reads_text, len(reads_text), subsample, C.byref(rev),
                                  C.byref(idx))
    if rc:
        raise OracleError(rc)
    return bool(rev.value), idx.value


E_NOTAB, E_DUPKEY, E_NOGENE = -8, -9, -10


def _blob(strings):
    return b"".join((x if isinstance(x, bytes) else x.encode()) + b"\0" for x in strings)


def generate_sample_names(paths):
    """utils.rs:18-49 → (names, fell_back_to_Sample_N)"""
    out = C.create_string_buffer(1 << 16)
    fb = C.c_int(0)
    rc = lib().orc_generate_sample_names(_blob(paths), len(paths), out, len(out), C.byref(fb))
    if rc:
        raise OracleError(rc)
    return (out.value.decode().split("\n") if paths else []), bool(fb.value)


class GeneMap:
    """genemap.rs GeneMap built from text."""

    def __init__(self, text: bytes):
        err = C.c_int(0)
        self.h = lib().orc_genemap_from_text(text, len(text), C.byref(err))
        if not self.h:
            raise OracleError(err.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_genemap_free(self.h)
            self.h = None

    def get(self, sgrna: bytes):
        n = C.c_size_t(0)
        p = lib().orc_genemap_get(self.h, sgrna, len(sgrna), C.byref(n))
        return C.string_at(p, n.value) if p else None

    def missing_aliases(self, library: Library):
        i = lib().orc_genemap_missing(self.h, library.h)
        return None if i < 0 else library.ids()[i]


def generate_columns(names, with_genemap=False):
    out = C.create_string_buffer(1 << 16)
    rc = lib().orc_generate_columns(_blob(names), len(names), int(with_genemap), out, len(out))
    if rc:
        raise OracleError(rc)
    return out.value.decode()


def format_results(library: Library, counts_per_sample, names, genemap: GeneMap = None, include_zero=False):
    """results.rs write_results → text.  counts_per_sample: list of per-guide count lists (library order)."""
    n = library.n()
    flat = (C.c_uint64 * (n * len(counts_per_sample)))(*[c for row in counts_per_sample for c in row])
    cap = 64 + (n + 2) * (64 + 24 * max(1, len(counts_per_sample))) + sum(len(x) + 2 for x in names)
    out = C.create_string_buffer(cap)
    rc = lib().orc_format_results(library.h, flat, len(counts_per_sample), _blob(names), genemap.h if genemap else None,
                                  int(include_zero), out, cap)
    if rc < 0:
        raise OracleError(rc)
    return out.raw[:rc].decode()
