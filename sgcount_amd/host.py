"""Host-side mirror of the reference's count-path interface, over the C ABI.

Same names, argument meaning and error behaviour as noamteyssier/sgcount v0.1.35:

    Offset            src/offsetter.rs:10-34
    Library           src/library.rs:9-99
    Permuter          src/permutes.rs:36-58
    Counter           src/counter.rs:17-66, :71-76, :239-251
    initialize_reader fxread (call sites src/count.rs:24,64,87)

so that the parity tests read like the reference's own tests.  All matching and counting runs
on the device through libsgcount_hip.so; nothing here falls back to a CPU implementation.
"""
import ctypes as C
import gzip
import weakref
from dataclasses import dataclass

import numpy as np

from . import _ffi


# ---------------------------------------------------------------------------------------------
# Offset — src/offsetter.rs:10-34
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Offset:
    kind: str   # "Forward" | "Reverse"
    idx: int

    @staticmethod
    def Forward(index):
        return Offset("Forward", int(index))

    @staticmethod
    def Reverse(index):
        return Offset("Reverse", int(index))

    def index(self):
        return self.idx

    def is_forward(self):
        return self.kind == "Forward"

    def is_reverse(self):
        return self.kind == "Reverse"

    def __repr__(self):
        return f"{self.kind}({self.idx})"


# ---------------------------------------------------------------------------------------------
# FASTX records — stand-in for fxread's reader (single-line FASTA, 4-line FASTQ, optional .gz)
# ---------------------------------------------------------------------------------------------
@dataclass
class Record:
    _id: bytes
    _seq: bytes

    def id(self):
        return self._id

    def seq(self):
        return self._seq


def parse_fastx(text: bytes):
    """Yields Records from FASTA ('>' two-line) or FASTQ ('@' four-line) text.  The reader decisions of DESIGN.md §2 (fxread is not
    pinned here): a '\r' before the '\n' belongs to the terminator; blank lines at the very end are not records; a FASTQ stream
    that ends behind a separator line ends with a record whose quality line is empty (written "+\n\n", "+\n" or "+")."""
    if not text:
        return
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    lines = [ln[:-1] if ln.endswith(b"\r") else ln for ln in lines]      # CRLF: '\r' belongs to the terminator (unpinned upstream)
    if text[:1] == b">":
        step, marker = 2, b">"
    elif text[:1] == b"@":
        step, marker = 4, b"@"
    else:
        raise ValueError("not FASTA/FASTQ: first byte is %r" % text[:1])
    i, n = 0, len(lines)
    while i < n:
        if lines[i] == b"":
            if any(lines[i:]):
                raise ValueError("malformed FASTX header at line %d" % (i + 1))
            return                                  # only blank lines are left
        if lines[i][:1] != marker:
            raise ValueError("malformed FASTX header at line %d" % (i + 1))
        if i + 1 >= n:
            raise ValueError("truncated FASTX record")
        if step == 4:
            if i + 2 >= n or lines[i + 2][:1] != b"+":
                raise ValueError("malformed FASTQ record at line %d" % (i + 3))
            # (i + 3 >= n: the stream ends behind the separator line — an empty quality line)
        yield Record(lines[i][1:], lines[i + 1])
        i += step


def read_path(path: str) -> bytes:
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def initialize_reader(path: str):
    """fxread::initialize_reader: path → iterator of Records (.gz by suffix)."""
    return parse_fastx(read_path(path))


# ---------------------------------------------------------------------------------------------
# Library — src/library.rs
# ---------------------------------------------------------------------------------------------
class Library:
    def __init__(self, table, order):
        self.table = table          # seq -> id            (src/library.rs:10)
        self._order = order         # sequences in file order (ours; the reference iterates HashMap order)
        sizes = {len(k) for k in table}
        if not table:
            raise RuntimeError("called `Option::unwrap()` on a `None` value")   # src/library.rs:74
        if len(sizes) != 1:
            raise ValueError("Library sequence sizes are inconsistent")         # src/library.rs:83
        self._size = sizes.pop()
        self._devices = {}

    @classmethod
    def from_reader(cls, reader):
        """src/library.rs:17-21 + :89-99 (panics on a duplicate sequence)."""
        table, order = {}, []
        for rec in reader:
            if rec.seq() in table:
                raise RuntimeError("Unexpected duplicate sequence in library found: %s" % rec.seq().decode())
            table[rec.seq()] = rec.id()
            order.append(rec.seq())
        return cls(table, order)

    @classmethod
    def from_hashmap(cls, table):
        return cls(dict(table), list(table))

    def contains(self, token):      # src/library.rs:34-40 (host-side dictionary view of the same table)
        return self.table.get(bytes(token))

    def alias(self, token):         # src/library.rs:44-46
        return self.table.get(bytes(token))

    def keys(self):                 # src/library.rs:49
        return iter(self._order)

    def values(self):               # src/library.rs:54
        return (self.table[k] for k in self._order)

    def size(self):                 # src/library.rs:60
        return self._size

    def __len__(self):
        return len(self._order)

    # -- device residency ------------------------------------------------------------------
    def device(self, one_mismatch: bool, device_index: int = 0, options=None):
        """The resident (library [+ permute]) tables for this library on one GPU.  options: sgc_set_option pairs applied
        before the tables are built (first call only)."""
        key = (bool(one_mismatch), device_index, tuple(sorted((options or {}).items())))
        dl = self._devices.get(key)
        if dl is None:
            dl = DeviceLibrary(self, one_mismatch, device_index, options)
            self._devices[key] = dl
        return dl


class DeviceLibrary:
    """sgc_ctx + sgc_set_library: what (&Library, &Option<Permuter>) is to Counter::new."""

    def __init__(self, library: Library, one_mismatch: bool, device_index: int = 0, options=None):
        self.lib = _ffi.load()
        self.library = library
        self.one_mismatch = bool(one_mismatch)
        self.ctx = C.c_void_p()
        _ffi.check(self.lib.sgc_init(device_index, C.byref(self.ctx)))
        self._fin = weakref.finalize(self, self.lib.sgc_free, self.ctx)
        for k, v in (options or {}).items():
            _ffi.check(self.lib.sgc_set_option(self.ctx, k.encode(), int(v)))
        flat = b"".join(library.keys())
        try:
            _ffi.check(self.lib.sgc_set_library(self.ctx, flat, len(library), library.size(), int(one_mismatch)))
        except _ffi.SgcError as e:
            if e.code == _ffi.E_DUPLICATE:
                raise RuntimeError(str(e)) from e
            raise
        # 8 / 16, or 0: a library of arbitrary bytes or L > 30 has no packed record format (byte-string path: reads or FASTQ text only)
        self.record_bytes = self.info().record_bytes

    def info(self):
        out = _ffi.LibInfo()
        _ffi.check(self.lib.sgc_library_info(self.ctx, C.byref(out)))
        return out

    def set_stream(self, stream_ptr):
        _ffi.check(self.lib.sgc_set_stream(self.ctx, C.c_void_p(stream_ptr)))

    def set_option(self, key, value):
        _ffi.check(self.lib.sgc_set_option(self.ctx, key.encode(), int(value)))

    def lookup(self, tokens, which=2):
        """gid (library order) or -1 per token; which: 0 library, 1 permuter, 2 library-then-permuter."""
        L = self.library.size()
        toks = [bytes(t) for t in tokens]
        bad = [i for i, t in enumerate(toks) if len(t) != L]
        flat = b"".join(t if len(t) == L else b"#" * L for t in toks)
        out = np.empty(len(toks), dtype=np.int32)
        _ffi.check(self.lib.sgc_lookup(self.ctx, flat, len(toks), which, out.ctypes.data))
        for i in bad:
            out[i] = -1
        return out

    def timing(self, enable=None, reset=False):
        if enable is not None:
            _ffi.check(self.lib.sgc_timing_enable(self.ctx, int(enable)))
            return None
        t = _ffi.Timing()
        _ffi.check(self.lib.sgc_timing_read(self.ctx, C.byref(t), int(reset)))
        return t


# ---------------------------------------------------------------------------------------------
# Permuter — src/permutes.rs
# ---------------------------------------------------------------------------------------------
class Permuter:
    """Unambiguous one-off table.  Built on the device by sgc_set_library(enable_1mm=1); this object
    is the handle Counter.new takes in place of `&Option<Permuter>`."""

    def __init__(self, sequences):
        seqs = list(sequences)
        self._library = Library.from_hashmap({s: b"%d" % i for i, s in enumerate(seqs)})
        self._dev = None

    @classmethod
    def new(cls, sequences):        # src/permutes.rs:47-50
        return cls(sequences)

    def _device(self):
        if self._dev is None:
            self._dev = self._library.device(True)
        return self._dev

    def contains(self, token):      # src/permutes.rs:55-57: child -> parent sequence
        g = int(self._device().lookup([token], which=1)[0])
        return None if g < 0 else self._library._order[g]


# ---------------------------------------------------------------------------------------------
# Counter — src/counter.rs
# ---------------------------------------------------------------------------------------------
def _flatten(seqs):
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    return b"".join(seqs), offsets


def pack_reads_host(seqs, L, offset: Offset, position_recursion: bool):
    """sgc_pack_reads_host over a list of read sequences → numpy array of records (u64, 1 or 2 per read)."""
    lib = _ffi.load()
    rb = lib.sgc_record_bytes(L)
    if not rb:
        raise _ffi.SgcError(_ffi.E_UNSUPPORTED, "guide length outside 1..30")
    flat, offs = _flatten(seqs)
    out = np.empty(len(seqs) * (rb // 8), dtype=np.uint64)
    _ffi.check(lib.sgc_pack_reads_host(flat, offs.ctypes.data, len(seqs), L, int(offset.is_reverse()), offset.index(),
                                       int(position_recursion), out.ctypes.data))
    return out


class Counter:
    def __init__(self):
        self.results = {}
        self._total = 0
        self._matched = 0
        self._guide_counts = None

    @classmethod
    def from_hashmap(cls, table):   # src/counter.rs:24-30
        c = cls()
        c.results = dict(table)
        return c

    @classmethod
    def new(cls, reader, library: Library, permuter, offset: Offset, size: int, position_recursion: bool,
            pack: str = "host", batch: int = 1 << 20, device_index: int = 0, options=None):
        """Counter::new (src/counter.rs:36-66): consumes `reader`, returns the finished Counter.
        pack = "host" (sgc_pack_reads_host, the north-star split), "device" (pack kernel), "windows" (the pack kernel on the
        pieces of the reads that hold their windows) or "fastq" (the reads as FASTQ text through sgc_sample_push_fastq).  A library without a packed record format (non-ACGT bytes, L > 30) is
        served from the read bytes whatever `pack` says — on the device either way."""
        if size != library.size():
            raise ValueError("size must equal library.size() (src/count.rs:31)")
        dev = library.device(permuter is not None, device_index, options)
        if pack == "host" and dev.record_bytes == 0:
            pack = "device"
        lib = dev.lib
        sample = C.c_void_p()
        _ffi.check(lib.sgc_sample_begin(dev.ctx, C.byref(sample), int(offset.is_reverse()), offset.index(),
                                        int(position_recursion)))
        try:
            chunk = []

            def flush():
                if not chunk:
                    return
                if pack == "host":
                    recs = pack_reads_host(chunk, size, offset, position_recursion)
                    _ffi.check(lib.sgc_sample_push_packed(sample, recs.ctypes.data, len(chunk), _ffi.MEM_HOST))
                elif pack == "device":
                    flat, offs = _flatten(chunk)
                    _ffi.check(lib.sgc_sample_push_reads(sample, flat, offs.ctypes.data, len(chunk), _ffi.MEM_HOST))
                elif pack == "windows":
                    # of every read only the piece that holds its windows (sgc_sample_push_windows), as the C++ scanner ships the
                    # reads it routes to the byte-string chain: oriented bases [max(o - 1, 0), min(len, o + L + 1))
                    o, rev = offset.index(), offset.is_reverse()
                    pieces = []
                    for r in chunk:
                        lo, hi = max(o - 1, 0), min(len(r), o + size + 1)
                        pieces.append(b"" if hi <= lo else (r[len(r) - hi: len(r) - lo] if rev else r[lo:hi]))
                    flat, offs = _flatten(pieces)
                    _ffi.check(lib.sgc_sample_push_windows(sample, flat, offs.ctypes.data, len(pieces), _ffi.MEM_HOST, 1 if o >= 1 else 0))
                elif pack == "fastq":
                    text = b"".join(b"@r\n%s\n+\n%s\n" % (r, b"I" * len(r)) for r in chunk)
                    _ffi.check(lib.sgc_sample_push_fastq(sample, text, len(text), _ffi.MEM_HOST, None))
                else:
                    raise ValueError("pack must be 'host', 'device', 'windows' or 'fastq'")
                _ffi.check(lib.sgc_sample_sync(sample))
                chunk.clear()

            for rec in reader:
                chunk.append(rec.seq())
                if len(chunk) >= batch:
                    flush()
            flush()
            counts = np.zeros(len(library), dtype=np.uint64)
            tot, mat = C.c_uint64(0), C.c_uint64(0)
            _ffi.check(lib.sgc_sample_finish(sample, counts.ctypes.data, C.byref(tot), C.byref(mat)))
        finally:
            lib.sgc_sample_free(sample)
        c = cls()
        c._guide_counts = counts
        c._total, c._matched = tot.value, mat.value
        # id-keyed fold (src/counter.rs:232-235): guides sharing an id pool their counts
        for seq, n in zip(library.keys(), counts.tolist()):
            if n:
                ident = library.table[seq]
                c.results[ident] = c.results.get(ident, 0) + n
        return c

    def get_value(self, token):     # src/counter.rs:71-76
        return self.results.get(bytes(token), 0)

    def total_reads(self):          # src/counter.rs:239
        return self._total

    def matched_reads(self):        # src/counter.rs:244
        return self._matched

    def fraction_mapped(self):      # src/counter.rs:249-251 (NaN for 0/0 like f64 division)
        return self._matched / self._total if self._total else float("nan")

    def guide_counts(self):
        """Per-guide counts in library order, before id pooling (device view)."""
        return self._guide_counts
