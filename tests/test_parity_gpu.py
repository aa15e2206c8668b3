"""Parity tests proper: the HIP path (through the C ABI, via the host mirror) against the reference's
known answers, the committed golden vectors and the CPU oracle.  Bit-exact: this is integer work.

Run on the GPU box with:  python -m pytest tests -m gpu
"""
import collections
import json
import os
import random

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import sgcount_amd
    sgcount_amd._ffi.load()
    return sgcount_amd


def _lib(S, text):
    return S.Library.from_reader(S.parse_fastx(text))


def _fasta(seqs, prefix=b"g"):
    return b"".join(b">%s%d\n%s\n" % (prefix, i, s) for i, s in enumerate(seqs))


def _reads_fasta(reads):
    return b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(reads))


# ---- the reference's own unit tests, through the ABI --------------------------------------------
LIB_ACTG = b">seq.0\nACTG\n"


@pytest.mark.parametrize("pack", ["host", "device"])
def test_counter_kats(S, pack):
    """counter.rs:283-320"""
    lib = _lib(S, LIB_ACTG)
    perm = S.Permuter.new(lib.keys())

    def run(read, p):
        return S.Counter.new(S.parse_fastx(b">seq.0\n" + read + b"\n"), lib, p, S.Offset.Forward(0), 4, False, pack=pack)

    assert run(b"ACTG", None).get_value(b"seq.0") == 1          # count_no_distance_no_permute
    assert run(b"AGTG", None).get_value(b"seq.0") == 0          # count_no_distance_with_permute / with_distance_no_permute
    c = run(b"AGTG", perm)                                       # count_with_distance_with_permute
    assert c.get_value(b"seq.0") == 1 and c.total_reads() == 1 and c.matched_reads() == 1


def test_permuter_kats(S):
    """permutes.rs:193-253 via sgc_lookup(which=1): child → unique parent"""
    p = S.Permuter.new([b"ACTG"])
    truth = [b"AATG", b"ACGG", b"ACAG", b"TCTG", b"GCTG", b"AGTG", b"ACTC", b"ATTG", b"ACCG", b"ACTT", b"CCTG", b"ACTA"]
    assert all(p.contains(t) == b"ACTG" for t in truth)
    assert p.contains(b"ACTG") is None                          # parents live in `null`
    p = S.Permuter.new([b"AC", b"CG"])
    for t in (b"GC", b"TC", b"AA", b"AT"):
        assert p.contains(t) == b"AC"
    for t in (b"CA", b"CT", b"GG", b"TG"):
        assert p.contains(t) == b"CG"
    for t in (b"AG", b"CG", b"CC", b"AC"):
        assert p.contains(t) is None
    assert p._device().info().perm_entries == 8                 # the 12 of permutes.rs:215 minus the 4 'N' children


def test_permuter_n_children_via_counter(S):
    """permutes.rs:196-199 'N' children (ACNG, NCTG, ANTG, ACTN → ACTG) are resolved in-kernel"""
    lib = _lib(S, LIB_ACTG)
    perm = S.Permuter.new(lib.keys())
    for child in (b"ACNG", b"NCTG", b"ANTG", b"ACTN"):
        c = S.Counter.new(S.parse_fastx(b">r\n" + child + b"\n"), lib, perm, S.Offset.Forward(0), 4, False)
        assert c.get_value(b"seq.0") == 1, child
    lib2 = _lib(S, b">a\nAC\n>b\nCG\n")
    perm2 = S.Permuter.new(lib2.keys())
    want = {b"NC": b"a", b"AN": b"a", b"CN": b"b", b"NG": b"b"}
    for child, ident in want.items():
        c = S.Counter.new(S.parse_fastx(b">r\n" + child + b"\n"), lib2, perm2, S.Offset.Forward(0), 2, False)
        assert c.get_value(ident) == 1 and c.matched_reads() == 1, child


def test_library_kats(S):
    """library.rs:119-136"""
    lib = _lib(S, LIB_ACTG)
    assert lib.size() == 4 and len(list(lib.keys())) == 1
    dev = lib.device(False)
    assert dev.lookup([b"ACTG", b"ACTT"], which=0).tolist() == [0, -1]
    with pytest.raises(RuntimeError):
        _lib(S, b">seq.0\nACTG\n>seq.1\nACTG\n")
    # the ABI itself reports duplicates too (SGC_E_DUPLICATE)
    bad = S.Library({b"ACTG": b"x"}, [b"ACTG", b"ACTG"])
    with pytest.raises(RuntimeError):
        bad.device(False)


def test_unsupported_library_fails_loudly(S):
    """non-ACGT bytes and L > 30 are served by the byte-string path (tests/test_generic_gpu.py); what is left is refused"""
    ffi = S._ffi
    with pytest.raises(ffi.SgcError) as e:
        _lib(S, b">a\n" + b"A" * 65536 + b"\n").device(False)
    assert e.value.code == ffi.E_UNSUPPORTED
    with pytest.raises(ffi.SgcError) as e:
        S.pack_reads_host([b"ACGT"], 31, S.Offset.Forward(0), True)
    assert e.value.code == ffi.E_UNSUPPORTED


def test_option_values_are_checked(S):
    """the retired variants (0: direct atomics, 2: the superseded lookup kernel) and slice sizes the kernels do not have are refused"""
    ffi = S._ffi
    dl = _lib(S, b">a\nACGTACGTACGTACGTACGT\n>b\nTTGTACGTACGTACGTACGA\n").device(True)
    for key, bad in (("variant", 0), ("variant", 2), ("variant", 5), ("slice_log2", 11), ("slice_log2", 14)):
        with pytest.raises(ffi.SgcError) as e:
            dl.set_option(key, bad)
        assert e.value.code == ffi.E_ARG, (key, bad)
    for key, ok in (("variant", 1), ("variant", 3), ("variant", 4), ("slice_log2", 12), ("slice_log2", 13), ("slice_log2", 0), ("wide", 0), ("wide", 1)):
        dl.set_option(key, ok)


# ---- example/ fixtures vs committed golden tables -------------------------------------------------
@pytest.mark.parametrize("pack", ["host", "device"])
def test_example_fixtures_golden(S, pack, example_library_text, example_reads):
    g = json.load(open(os.path.join(GOLDEN, "example_counts.json")))
    lib = _lib(S, example_library_text)
    perm = S.Permuter.new(lib.keys())
    for c in g["cases"]:
        reads = example_reads[c["file"][:-9]]
        ctr = S.Counter.new(S.parse_fastx(reads), lib, None if c["exact"] else perm, S.Offset.Forward(g["offset"]),
                            lib.size(), c["position_recursion"], pack=pack)
        assert ctr.guide_counts().tolist() == c["counts"], c["file"]
        assert ctr.total_reads() == c["total"] and ctr.matched_reads() == c["matched"]
        assert [ctr.get_value(i) for i in lib.values()] == c["counts"]


@pytest.mark.parametrize("pack", ["host", "device"])
def test_edge_cases_golden(S, pack):
    """Hand-made and seeded cases: N / lowercase / IUPAC bytes, short and empty reads, offset 0,
    reverse strand, ambiguity, REC8 max length (23) and REC16 lengths (24, 30)."""
    g = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))
    libs = {}
    for c in g["cases"]:
        key = tuple(c["guides"])
        if key not in libs:
            libs[key] = _lib(S, _fasta([s.encode() for s in c["guides"]]))
        lib = libs[key]
        perm = None if c["exact"] else S.Permuter.new(lib.keys())
        off = S.Offset.Reverse(c["offset"]) if c["reverse"] else S.Offset.Forward(c["offset"])
        reads = [S.Record(b"r", r.encode("latin1")) for r in c["reads"]]
        ctr = S.Counter.new(iter(reads), lib, perm, off, lib.size(), c["position_recursion"], pack=pack)
        want = collections.Counter(a for a in c["assign"] if a >= 0)
        got = {i: v for i, v in enumerate(ctr.guide_counts().tolist()) if v}
        assert got == dict(want), (c["name"], c["exact"], c["position_recursion"])
        assert ctr.total_reads() == len(reads) and ctr.matched_reads() == sum(want.values())


def test_edge_cases_per_read(S):
    """Each golden read on its own: pins the per-read assignment, not only the totals."""
    g = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))
    for c in g["cases"]:
        if not c["name"].startswith("handmade"):
            continue
        lib = _lib(S, _fasta([s.encode() for s in c["guides"]]))
        perm = None if c["exact"] else S.Permuter.new(lib.keys())
        off = S.Offset.Reverse(c["offset"]) if c["reverse"] else S.Offset.Forward(c["offset"])
        for r, a in zip(c["reads"], c["assign"]):
            ctr = S.Counter.new(iter([S.Record(b"r", r.encode("latin1"))]), lib, perm, off, lib.size(),
                                c["position_recursion"])
            got = [i for i, v in enumerate(ctr.guide_counts().tolist()) if v]
            assert got == ([a] if a >= 0 else []), (c["name"], r, c["exact"], c["position_recursion"])


# ---- seeded random inputs vs the oracle ------------------------------------------------------------
def _random_case(rng, L, n_guides, n_reads, o):
    alpha = b"ACGT"
    guides, seen = [], set()
    while len(guides) < n_guides:
        if guides and rng.random() < 0.15:        # plant Hamming-1/2 neighbours
            s = bytearray(rng.choice(guides))
            for _ in range(rng.choice([1, 2])):
                s[rng.randrange(L)] = rng.choice(alpha)
            s = bytes(s)
        else:
            s = bytes(rng.choice(alpha) for _ in range(L))
        if s not in seen:
            seen.add(s); guides.append(s)
    reads = []
    for _ in range(n_reads):
        g = bytearray(rng.choice(guides))
        u = rng.random()
        if u < 0.2:
            g[rng.randrange(L)] = rng.choice(b"ACGTN")
        elif u < 0.25:
            g[rng.randrange(L)] = rng.choice(b"ACGTN"); g[rng.randrange(L)] = rng.choice(b"ACGTNa")
        elif u < 0.3:
            g = bytearray(rng.choice(alpha) for _ in range(L))
        pre_len = o + rng.choice([0, 0, 0, 1, -1, 2])
        pre = bytes(rng.choice(alpha) for _ in range(max(pre_len, 0)))
        tail = bytes(rng.choice(alpha) for _ in range(rng.choice([0, 1, 2, 5, 30])))
        r = pre + bytes(g) + tail
        if rng.random() < 0.03:
            r = r[: rng.randrange(len(r) + 1)]
        reads.append(r)
    return guides, reads


@pytest.mark.parametrize("o", [0, 1, 2, 31])
@pytest.mark.parametrize("reverse", [False, True])
def test_window_pieces_count_like_whole_reads(S, o, reverse):
    """sgc_sample_push_windows: of every read only the piece that holds its windows (what the C++ scanner ships for the reads it
    routes to the byte-string chain of a hybrid library).  Same table as the oracle on the whole reads, for offsets at and next
    to zero, reads too short for some or all windows, both strands, with and without the position recursion."""
    rng = random.Random(77 + o + 10 * reverse)
    L = 20
    guides, reads = _random_case(rng, L, 800, 12000, o)
    reads += [b"", b"A", b"ACGT" * 3, bytes(guides[0])[: L - 1], bytes(guides[1])]
    if reverse:
        reads = [bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r)) for r in reads]
    lib_text, reads_text = _fasta(guides), _reads_fasta(reads)
    lib = _lib(S, lib_text)
    perm = S.Permuter.new(lib.keys())
    off = S.Offset.Reverse(o) if reverse else S.Offset.Forward(o)
    for exact in (False, True):
        for recursion in (True, False):
            want, tot, mat = O.count_text(lib_text, reads_text, reverse, o, exact, recursion)
            ctr = S.Counter.new(S.parse_fastx(reads_text), lib, None if exact else perm, off, L, recursion, pack="windows", batch=5000)
            assert ctr.guide_counts().tolist() == want, (exact, recursion)
            assert (ctr.total_reads(), ctr.matched_reads()) == (tot, mat)


@pytest.mark.parametrize("variant", [4, 3])
@pytest.mark.parametrize("L,n_guides", [(20, 2000), (12, 300), (23, 500), (27, 400)])
@pytest.mark.parametrize("reverse", [False, True])
def test_random_vs_oracle(S, L, n_guides, reverse, variant):
    """variant 4 = in-LDS core resolver (the default), 3 = probing resolver; both must give the oracle's table"""
    rng = random.Random(1000 * L + n_guides + reverse)
    o = 9
    guides, reads = _random_case(rng, L, n_guides, 20000, o)
    if reverse:
        reads = [bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r)) if rng.random() < 0.97 else r
                 for r in reads]
    lib_text, reads_text = _fasta(guides), _reads_fasta(reads)
    lib = _lib(S, lib_text)
    perm = S.Permuter.new(lib.keys())
    off = S.Offset.Reverse(o) if reverse else S.Offset.Forward(o)
    for exact in (False, True):
        lib.device(not exact).set_option("variant", variant)
        for recursion in (True, False):
            want, tot, mat = O.count_text(lib_text, reads_text, reverse, o, exact, recursion)
            for pack in ("host", "device", "windows"):
                ctr = S.Counter.new(S.parse_fastx(reads_text), lib, None if exact else perm, off, L, recursion,
                                    pack=pack, batch=7001)
                assert ctr.guide_counts().tolist() == want, (exact, recursion, pack)
                assert (ctr.total_reads(), ctr.matched_reads()) == (tot, mat)
                assert int(ctr.guide_counts().sum()) == mat


def _dense_case(rng, L, o):
    """Families built to stress the single-mismatch rules: siblings (guides one substitution apart: their
    shared children are ambiguous), cousins (two apart: the child in between has two parents), and reads that
    walk every substitution / 'N' at every position, at the three alignments, alone and combined."""
    alpha = b"ACGT"
    guides, seen = [], set()

    def add(s):
        s = bytes(s)
        if s not in seen:
            seen.add(s); guides.append(s)
    for _ in range(12):
        g = bytearray(rng.choice(alpha) for _ in range(L))
        add(g)
        for _ in range(3):                       # siblings and cousins
            h = bytearray(g)
            for _ in range(rng.choice([1, 2])):
                h[rng.randrange(L)] = rng.choice(alpha)
            add(h)
        h = bytearray(g[1:] + bytes([rng.choice(alpha)]))      # the same guide shifted by one base
        add(h)
    while len(guides) < 200:
        add(bytearray(rng.choice(alpha) for _ in range(L)))
    reads = []
    for g in guides[:70]:
        for j in range(L):
            for b in b"ACGTN":
                w = bytearray(g); w[j] = b
                for shift in (0, 1, -1):
                    pre = bytes(rng.choice(alpha) for _ in range(o + shift))
                    reads.append(pre + bytes(w) + bytes(rng.choice(alpha) for _ in range(3)))
    for _ in range(4000):                        # two edits, 'N' + substitution, junk
        w = bytearray(rng.choice(guides))
        w[rng.randrange(L)] = rng.choice(b"ACGTN"); w[rng.randrange(L)] = rng.choice(b"ACGTN")
        pre = bytes(rng.choice(b"ACGTN") for _ in range(o + rng.choice([0, 1, -1])))
        reads.append(pre + bytes(w) + bytes(rng.choice(b"ACGTN") for _ in range(3)))
    rng.shuffle(reads)
    return guides, reads


@pytest.mark.parametrize("L", [20, 23, 9, 6, 4])
def test_dense_neighbourhoods_vs_oracle(S, L):
    """Every substitution and every 'N' of whole guide families, at all three alignments: the in-LDS core
    resolver (variant 4), the probing resolver (variant 3) and the oracle agree read for read in aggregate."""
    rng = random.Random(77 + L)
    o = 5
    guides, reads = _dense_case(rng, L, o)
    lib_text, reads_text = _fasta(guides), _reads_fasta(reads)
    for recursion in (True, False):
        want, tot, mat = O.count_text(lib_text, reads_text, False, o, False, recursion)
        for variant in (4, 3):
            lib = _lib(S, lib_text)
            lib.device(True).set_option("variant", variant)
            perm = S.Permuter.new(lib.keys())
            ctr = S.Counter.new(S.parse_fastx(reads_text), lib, perm, S.Offset.Forward(o), L, recursion, pack="device")
            assert ctr.guide_counts().tolist() == want, (variant, recursion)
            assert (ctr.total_reads(), ctr.matched_reads()) == (tot, mat)


def test_core_index_overflow_falls_back(S):
    """3000 guides sharing their first 12 bases: one core value carries more entries than a core partition may
    hold, so no core index is built and the probing resolver (variant 3) serves the library — same table."""
    rng = random.Random(99)
    L, o, head = 20, 5, b"ACGTTGCAACGT"
    guides, seen = [], set()
    while len(guides) < 3000:
        g = head + bytes(rng.choice(b"ACGT") for _ in range(L - len(head)))
        if g not in seen:
            seen.add(g); guides.append(g)
    reads = []
    for _ in range(20000):
        w = bytearray(rng.choice(guides))
        if rng.random() < 0.5:
            w[rng.randrange(L)] = rng.choice(b"ACGTN")
        pre = bytes(rng.choice(b"ACGT") for _ in range(o + rng.choice([0, 0, 1, -1])))
        reads.append(pre + bytes(w) + b"GATTACA")
    lib_text, reads_text = _fasta(guides), _reads_fasta(reads)
    want, tot, mat = O.count_text(lib_text, reads_text, False, o, False, True)
    lib = _lib(S, lib_text)
    info = lib.device(True).info()
    assert info.core_partitions == 0 and info.path == 3 and info.slices >= 1         # the fallback is visible to the host
    ctr = S.Counter.new(S.parse_fastx(reads_text), lib, S.Permuter.new(lib.keys()), S.Offset.Forward(o), L, True, pack="device")
    assert ctr.guide_counts().tolist() == want and (ctr.total_reads(), ctr.matched_reads()) == (tot, mat)
    # an ordinary library of the same size gets its core index
    other = _lib(S, _fasta(_random_case(random.Random(3), L, 3000, 1, o)[0]))
    info = other.device(True).info()
    assert info.core_partitions >= 1 and info.path == 4


def test_device_build_matches_host_build(S):
    """The single-mismatch table built on the GPU (children -> rocPRIM sort -> CAS inserts, sgc_build.hip) answers
    every probe like the host-built one (sgc_tables.cpp) and like the oracle's Permuter, and holds as many entries."""
    rng = random.Random(2024)
    guides, _ = _random_case(rng, 14, 400, 1, 3)
    lib_text = _fasta(guides)
    probes = set(guides)
    for g in guides[:120]:
        for j in range(14):
            for b in b"ACGT":
                w = bytearray(g); w[j] = b
                probes.add(bytes(w))
    for _ in range(3000):
        probes.add(bytes(rng.choice(b"ACGT") for _ in range(14)))
    probes = sorted(probes)
    olib = O.Library(lib_text)
    operm = O.Permuter(olib)
    index = {g: i for i, g in enumerate(guides)}
    want = []
    for w in probes:
        parent = operm.contains(w) if w not in index else None      # the permuter is consulted after the library (counter.rs:113)
        want.append(index[parent] if parent is not None else -1)
    got, entries = {}, {}
    for mode in ("device", "host"):
        lib = _lib(S, lib_text)
        dev = lib.device(True, options={"host_build": int(mode == "host")})
        out = dev.lookup(probes, which=1).tolist()
        got[mode] = [(-1 if w in index else v) for w, v in zip(probes, out)]     # children that are guides are unreachable
        entries[mode] = dev.info().perm_entries
    assert got["device"] == got["host"] == want
    assert entries["device"] == entries["host"] > 0


def test_duplicate_ids_pool_counts(S):
    """counter.rs:232-235 folds by id: two guides sharing an id report the pooled count on both rows"""
    lib_text = b">same\nACGTAC\n>same\nTTGCAA\n>other\nCCCCCC\n"
    reads = _reads_fasta([b"ACGTAC", b"ACGTAC", b"TTGCAA", b"CCCCCC"])
    lib = _lib(S, lib_text)
    ctr = S.Counter.new(S.parse_fastx(reads), lib, None, S.Offset.Forward(0), 6, False)
    want, _, _ = O.count_text(lib_text, reads, False, 0, True, False)
    assert [ctr.get_value(i) for i in lib.values()] == want == [3, 3, 1]


def test_empty_and_zero_reads(S):
    lib = _lib(S, LIB_ACTG)
    c = S.Counter.new(iter([]), lib, None, S.Offset.Forward(0), 4, True)
    assert c.total_reads() == 0 and c.matched_reads() == 0 and c.fraction_mapped() != c.fraction_mapped()
    c = S.Counter.new(iter([S.Record(b"r", b"")] * 5), lib, S.Permuter.new(lib.keys()), S.Offset.Forward(0), 4, True)
    assert c.total_reads() == 5 and c.matched_reads() == 0


def test_device_resident_push_and_accumulation(S):
    """Pushing device-resident records in several batches == one batch (linearity of the fold)."""
    import ctypes as C
    import torch
    rng = random.Random(5)
    guides, reads = _random_case(rng, 20, 500, 30000, 4)
    lib = _lib(S, _fasta(guides))
    dev = lib.device(True)
    ffi, L = S._ffi, dev.lib
    recs = S.pack_reads_host(reads, 20, S.Offset.Forward(4), True)
    d = torch.from_numpy(recs.view(np.int64)).cuda()
    torch.cuda.synchronize()

    def run(splits):
        smp = C.c_void_p()
        ffi.check(L.sgc_sample_begin(dev.ctx, C.byref(smp), 0, 4, 1))
        bounds = [0] + splits + [len(reads)]
        for a, b in zip(bounds[:-1], bounds[1:]):
            ffi.check(L.sgc_sample_push_packed(smp, d.data_ptr() + 8 * a, b - a, ffi.MEM_DEVICE))
        out = np.zeros(len(guides), dtype=np.uint64)
        t, m = C.c_uint64(), C.c_uint64()
        ffi.check(L.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m)))
        L.sgc_sample_free(smp)
        return out.tolist(), t.value, m.value

    one = run([])
    assert one == run([1, 64, 65, 10000, 29999])
    want, tot, mat = O.count_text(_fasta(guides), _reads_fasta(reads), False, 4, False, True)
    assert one == (want, tot, mat)


def test_async_packed_pushes_are_batched_on_the_device(S):
    """sgc_sample_push_packed_async: records pushed from pinned host memory in many small pieces are uploaded on the upload
    stream into two alternating device batch buffers and counted one batch at a time — same table as ONE synchronous push,
    whatever the batch size (smaller than a push, not a multiple of it, larger than everything), across a sample reset, and
    with a finish in the middle of a batch."""
    import ctypes as C
    rng = random.Random(11)
    guides, reads = _random_case(rng, 20, 400, 50000, 4)
    lib = _lib(S, _fasta(guides))
    dev = lib.device(True)
    ffi, L = S._ffi, dev.lib
    recs = S.pack_reads_host(reads, 20, S.Offset.Forward(4), True)
    n = len(reads)
    pinned = L.sgc_alloc_pinned(n * 8)
    assert pinned
    C.memmove(pinned, recs.ctypes.data, n * 8)
    want, tot, mat = O.count_text(_fasta(guides), _reads_fasta(reads), False, 4, False, True)

    def finish(smp):
        out = np.zeros(len(guides), dtype=np.uint64)
        t, m = C.c_uint64(), C.c_uint64()
        ffi.check(L.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m)))
        return out.tolist(), t.value, m.value

    try:
        for batch, piece in ((1000, 333), (4096, 4096), (7, 50000), (1 << 20, 1234), (20000, 20000)):
            ffi.check(L.sgc_set_option(dev.ctx, b"batch_records", batch))
            smp = C.c_void_p()
            ffi.check(L.sgc_sample_begin(dev.ctx, C.byref(smp), 0, 4, 1))
            for rep in range(2):                                   # the second round after a reset: the batch state starts over
                for a in range(0, n, piece):
                    m = min(piece, n - a)
                    ffi.check(L.sgc_sample_push_packed_async(smp, pinned + 8 * a, m))
                    if a == piece * 3:
                        assert finish(smp)[1] == a + m             # a finish in the middle of a batch counts what was pushed so far
                ffi.check(L.sgc_sample_wait_uploads(smp, 0))
                assert finish(smp) == (want, tot, mat), (batch, piece, rep)
                ffi.check(L.sgc_sample_reset(smp))
            L.sgc_sample_free(smp)
    finally:
        ffi.check(L.sgc_set_option(dev.ctx, b"batch_records", 1 << 24))
        L.sgc_free_pinned(pinned)


def test_cloned_context_shares_the_tables(S):
    """sgc_ctx_clone: a second ctx on the same device with its own stream and scratch but the SAME library tables (the reference
    lends one Library / Permuter to all its rayon workers, count.rs:103-136).  The clone counts the oracle's table, also after the
    source ctx has been freed, and a clone of a ctx without a library is refused."""
    import ctypes as C
    rng = random.Random(23)
    guides, reads = _random_case(rng, 20, 300, 20000, 4)
    lib = _lib(S, _fasta(guides))
    ffi = S._ffi
    L = ffi.load()
    want, tot, mat = O.count_text(_fasta(guides), _reads_fasta(reads), False, 4, False, True)
    recs = S.pack_reads_host(reads, 20, S.Offset.Forward(4), True)
    flat = b"".join(guides)
    src, bare, clone = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ffi.check(L.sgc_init(0, C.byref(bare)))
    assert L.sgc_ctx_clone(bare, C.byref(clone)) == ffi.E_STATE
    L.sgc_free(bare)
    ffi.check(L.sgc_init(0, C.byref(src)))
    ffi.check(L.sgc_set_library(src, flat, len(guides), 20, 1))
    ffi.check(L.sgc_ctx_clone(src, C.byref(clone)))

    def count(ctx):
        smp = C.c_void_p()
        ffi.check(L.sgc_sample_begin(ctx, C.byref(smp), 0, 4, 1))
        ffi.check(L.sgc_sample_push_packed(smp, recs.ctypes.data, len(reads), ffi.MEM_HOST))
        out = np.zeros(len(guides), dtype=np.uint64)
        t, m = C.c_uint64(), C.c_uint64()
        ffi.check(L.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m)))
        L.sgc_sample_free(smp)
        return out.tolist(), t.value, m.value

    assert count(clone) == (want, tot, mat) and count(src) == (want, tot, mat)
    info = ffi.LibInfo()
    ffi.check(L.sgc_library_info(clone, C.byref(info)))
    assert info.n_guides == len(guides) and info.one_mismatch == 1
    L.sgc_free(src)                                   # the tables live on with the clone
    assert count(clone) == (want, tot, mat)
    ffi.check(L.sgc_set_library(clone, flat, len(guides), 20, 0))      # a new library on the clone: the shared tables are released
    assert count(clone)[0] == O.count_text(_fasta(guides), _reads_fasta(reads), False, 4, True, True)[0]
    L.sgc_free(clone)
