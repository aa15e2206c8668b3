"""GPU: the byte-string count path (sgc_bytes.h / sgc_bytes.hip) — libraries the 2-bit records cannot represent.

Upstream compares raw bytes: any byte is a legal library symbol (src/library.rs:89-99), 'N' is the fifth letter of the
permute lexicon (src/permutes.rs:3) and a guide may have any length.  These tests drive such libraries (and, with the
"force_bytes" option, ordinary ones) through the C ABI and compare with the oracle, which is a byte-string restatement
and needs no special case for them."""
import collections
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN
from test_ingest_gpu import _count_parts, _cut_at_lines

pytestmark = pytest.mark.gpu

FORCE = {"force_bytes": 1}


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


def test_library_modes(S):
    ffi = S._ffi
    assert _lib(S, b">a\nACNG\n").device(True).record_bytes == 0                 # 'N' in the library
    assert _lib(S, b">a\n" + b"A" * 31 + b"\n").device(False).record_bytes == 0  # longer than one packed record
    assert _lib(S, b">a\nacgt\n").device(True).record_bytes == 0                 # lowercase is just another byte
    assert _lib(S, b">a\nACGT\n").device(True).record_bytes == 8
    dl = _lib(S, b">a\nACNG\n>b\nTTTT\n").device(True)
    # packed records cannot carry such a library: refused, not mis-counted
    smp = C.c_void_p()
    ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, 0, 1))
    rec = np.zeros(1, dtype=np.uint64)
    assert dl.lib.sgc_sample_push_packed(smp, rec.ctypes.data, 1, ffi.MEM_HOST) == ffi.E_STATE
    dl.lib.sgc_sample_free(smp)
    with pytest.raises(RuntimeError):                                             # duplicates panic as before (library.rs:91-96)
        S.Library({b"ACNG": b"x"}, [b"ACNG", b"ACNG"]).device(False)


def test_permuter_kats_with_n_children(S):
    """permutes.rs:193-253 — the full 16-children list of the singleton test, 'N' children included, and the 12-key map
    of {AC, CG}: on the byte path the 'N' children are table entries, as upstream."""
    lib = S.Library.from_hashmap({b"ACTG": b"0"})
    dl = lib.device(True, options=FORCE)
    truth = [b"AATG", b"ACGG", b"ACNG", b"ACAG", b"NCTG", b"TCTG", b"ANTG", b"GCTG", b"AGTG", b"ACTC", b"ATTG", b"ACCG",
             b"ACTT", b"ACTN", b"CCTG", b"ACTA"]
    assert dl.lookup(truth, which=1).tolist() == [0] * 16
    assert dl.lookup([b"ACTG"], which=1).tolist() == [-1]          # parents live in `null`
    assert dl.info().perm_entries == 16
    lib = S.Library.from_hashmap({b"AC": b"0", b"CG": b"1"})
    dl = lib.device(True, options=FORCE)
    assert dl.info().perm_entries == 12                            # permutes.rs:215
    for t, g in ((b"GC", 0), (b"TC", 0), (b"NC", 0), (b"AA", 0), (b"AT", 0), (b"AN", 0),
                 (b"CA", 1), (b"CT", 1), (b"CN", 1), (b"GG", 1), (b"TG", 1), (b"NG", 1)):
        assert dl.lookup([t], which=1).tolist() == [g], t
    for t in (b"AG", b"CG", b"CC", b"AC"):                         # the null set of permutes.rs:242-253
        assert dl.lookup([t], which=1).tolist() == [-1], t
    # against the oracle's Permuter, every 2-mer over ACGTN
    op = O.Permuter(seqs=[b"AC", b"CG"])
    for a in b"ACGTN":
        for b in b"ACGTN":
            t = bytes([a, b])
            want = op.contains(t)
            got = int(dl.lookup([t], which=1)[0])
            assert (list(lib.keys())[got] if got >= 0 else None) == want, t


@pytest.mark.parametrize("pack", ["device", "fastq"])
def test_edge_cases_golden_on_the_byte_path(S, pack):
    """the committed golden edge cases (tests/golden/edge_cases.json) with every library forced onto the byte path"""
    g = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))
    libs = {}
    for c in g["cases"]:
        if pack == "fastq" and any(("\n" in r or "\r" in r) for r in c["reads"]):
            continue
        key = tuple(c["guides"])
        if key not in libs:
            libs[key] = _lib(S, _fasta([s.encode() for s in c["guides"]]))
        lib = libs[key]
        perm = None if c["exact"] else S.Permuter.new(lib.keys())
        off = S.Offset.Reverse(c["offset"]) if c["reverse"] else S.Offset.Forward(c["offset"])
        reads = [S.Record(b"r", r.encode("latin1")) for r in c["reads"]]
        ctr = S.Counter.new(iter(reads), lib, perm, off, lib.size(), c["position_recursion"], pack=pack, options=FORCE)
        want = collections.Counter(a for a in c["assign"] if a >= 0)
        got = {i: v for i, v in enumerate(ctr.guide_counts().tolist()) if v}
        assert got == dict(want), (c["name"], c["exact"], c["position_recursion"])
        assert ctr.total_reads() == len(reads) and ctr.matched_reads() == sum(want.values())


def _random_case(rng, L, n_guides, n_reads, o, lib_alpha, read_alpha):
    guides, seen = [], set()
    while len(guides) < n_guides:
        if guides and rng.random() < 0.2:         # plant Hamming-1/2 neighbours: shared and nulled children
            s = bytearray(rng.choice(guides))
            for _ in range(rng.choice([1, 2])):
                s[rng.randrange(L)] = rng.choice(lib_alpha)
            s = bytes(s)
        else:
            s = bytes(rng.choice(lib_alpha) for _ in range(L))
        if s not in seen:
            seen.add(s); guides.append(s)
    reads = []
    for _ in range(n_reads):
        g = bytearray(rng.choice(guides))
        u = rng.random()
        if u < 0.3:
            g[rng.randrange(L)] = rng.choice(read_alpha)
        elif u < 0.4:
            g[rng.randrange(L)] = rng.choice(read_alpha); g[rng.randrange(L)] = rng.choice(read_alpha)
        elif u < 0.45:
            g = bytearray(rng.choice(read_alpha) for _ in range(L))
        pre = bytes(rng.choice(b"ACGT") for _ in range(max(o + rng.choice([0, 0, 0, 1, -1, 2]), 0)))
        tail = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 1, 2, 5, 30])))
        r = pre + bytes(g) + tail
        if rng.random() < 0.03:
            r = r[: rng.randrange(len(r) + 1)]
        reads.append(r)
    return guides, reads


@pytest.mark.parametrize("L,n_guides,lib_alpha", [
    (20, 1500, b"ACGTN"),          # 'N' inside guides: its children at that position are the four bases
    (12, 300, b"ACGTacgtN"),       # lowercase guides: distinct strings, children only ever carry ACGTN
    (31, 400, b"ACGT"),            # one base beyond the packed format
    (64, 200, b"ACGT"),
    (4, 40, b"ACGTN"),
    (1, 3, b"ACGTN"),
])
@pytest.mark.parametrize("reverse", [False, True])
def test_random_vs_oracle(S, L, n_guides, lib_alpha, reverse):
    rng = random.Random(77 * L + n_guides + reverse)
    o = 7
    guides, reads = _random_case(rng, L, n_guides, 12000, o, lib_alpha, b"ACGTNacgtRJ")
    if reverse:
        reads = [bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r)) if rng.random() < 0.97 else r for r in reads]
    lib_text, reads_text = _fasta(guides), _reads_fasta(reads)
    lib = _lib(S, lib_text)
    perm = S.Permuter.new(lib.keys())
    off = S.Offset.Reverse(o) if reverse else S.Offset.Forward(o)
    for exact in (False, True):
        for recursion in (True, False):
            want, tot, mat = O.count_text(lib_text, reads_text, reverse, o, exact, recursion)
            for pack in ("device", "fastq"):
                rs = [r for r in S.parse_fastx(reads_text)]
                ctr = S.Counter.new(iter(rs), lib, None if exact else perm, off, L, recursion, pack=pack, batch=5003, options=FORCE)
                assert ctr.guide_counts().tolist() == want, (exact, recursion, pack)
                assert (ctr.total_reads(), ctr.matched_reads()) == (tot, mat)


def test_acgt_library_same_table_on_both_paths(S):
    """an ordinary library counted by the packed path and, forced, by the byte path: the same table, and the oracle's"""
    rng = random.Random(3)
    guides, reads = _random_case(rng, 20, 3000, 60000, 11, b"ACGT", b"ACGTNa")
    lib_text, reads_text = _fasta(guides), _reads_fasta(reads)
    lib = _lib(S, lib_text)
    perm = S.Permuter.new(lib.keys())
    want, tot, mat = O.count_text(lib_text, reads_text, False, 11, False, True)
    a = S.Counter.new(S.parse_fastx(reads_text), lib, perm, S.Offset.Forward(11), 20, True, pack="device")
    b = S.Counter.new(S.parse_fastx(reads_text), lib, perm, S.Offset.Forward(11), 20, True, pack="device", options=FORCE)
    assert a.guide_counts().tolist() == want == b.guide_counts().tolist()
    assert (b.total_reads(), b.matched_reads()) == (tot, mat)


@pytest.mark.parametrize("reverse", [False, True])
def test_fastq_parts_on_the_byte_path(S, reverse):
    """the streaming ingest (parts at any line phase, host and device text, CRLF, no final newline, markers) with a
    library that needs the byte path"""
    import torch
    ffi = S._ffi
    rng = random.Random(21 + reverse)
    L, o = 33, 5
    guides, reads = _random_case(rng, L, 500, 20000, o, b"ACGTN", b"ACGTNa")
    reads[5] = b""
    reads[9] = b"ACGT"
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    lib_text = _fasta(guides)
    lib = _lib(S, lib_text)
    dl = lib.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    want = O.count_text(lib_text, text, reverse, o, False, True)
    assert _count_parts(torch, S, dl, [text], reverse, o, True, ffi.MEM_HOST) == want
    assert _count_parts(torch, S, dl, [text[:-1]], reverse, o, True, ffi.MEM_DEVICE) == want
    for n_cuts in (1, 3, 40):
        parts = _cut_at_lines(text, rng, n_cuts)
        assert _count_parts(torch, S, dl, parts, reverse, o, True, ffi.MEM_HOST) == want
        assert _count_parts(torch, S, dl, parts, reverse, o, True, ffi.MEM_DEVICE) == want
    assert _count_parts(torch, S, dl, _cut_at_lines(text, rng, 5), reverse, o, True, ffi.MEM_HOST, announce=False) == want
    crlf = text.replace(b"\n", b"\r\n")
    assert _count_parts(torch, S, dl, _cut_at_lines(crlf, rng, 7), reverse, o, True, ffi.MEM_HOST) == want
    # a part that ends with an unterminated SEQUENCE line (a truncated stream): the line is still a record
    cut = text.index(b"\n", text.index(b"@r100\n") + 6)
    head = text[:cut]
    want_head = O.count_text(lib_text, head + b"\n+\n\n", reverse, o, False, True)
    got = _count_parts(torch, S, dl, [head], reverse, o, True, ffi.MEM_HOST)
    assert got == want_head
    # marker bytes
    bad = bytearray(text)
    bad[text.index(b"@r77\n")] = ord("#")
    _count_parts(torch, S, dl, _cut_at_lines(bytes(bad), rng, 3), reverse, o, True, ffi.MEM_HOST, finish_rc=ffi.E_FORMAT)
    assert b"does not start with its marker byte" in dl.lib.sgc_last_error()


def _hybrid_library(rng, L, n_guides, frac_other):
    """mostly ACGT guides, a fraction with bytes outside ACGT (one 'N' mostly; two; a lowercase letter), and planted neighbours:
    ACGT guides that equal an 'N' guide up to that position (the shadow keys), pairs at distance 1 and 2 across both kinds"""
    guides, seen = [], set()

    def add(s):
        s = bytes(s)
        if s not in seen:
            seen.add(s); guides.append(s)
    while len(guides) < n_guides:
        u = rng.random()
        if guides and u < 0.25:
            s = bytearray(rng.choice(guides))                         # a neighbour of an existing guide (either kind)
            for _ in range(rng.choice([1, 1, 2])):
                s[rng.randrange(L)] = rng.choice(b"ACGT")
            add(s)
        elif u < 0.25 + frac_other:
            s = bytearray(rng.choice(b"ACGT") for _ in range(L))
            kind = rng.random()
            if kind < 0.7:
                s[rng.randrange(L)] = ord("N")
            elif kind < 0.85:
                s[rng.randrange(L)] = ord("N"); s[rng.randrange(L)] = ord("N")
            else:
                s[rng.randrange(L)] = rng.choice(b"acgtRY")
            add(s)
        else:
            add(bytearray(rng.choice(b"ACGT") for _ in range(L)))
    return guides


@pytest.mark.parametrize("L,n_guides,frac", [(20, 2000, 0.01), (20, 1500, 0.10), (12, 600, 0.30), (24, 800, 0.05), (6, 60, 0.2)])
@pytest.mark.parametrize("reverse", [False, True])
def test_hybrid_library_vs_oracle(S, L, n_guides, frac, reverse):
    """A library with a FEW guides outside ACGT ('N', lowercase): the packed pass serves the ACGT guides and only the reads a
    guide of the other kind could influence — a byte outside ACGT in the span region, or a window one substitution away from a
    single-'N' guide — take the byte-string chain over the whole library (sgc_bytes.hip, k_bytes_route).  Same table as the
    oracle (which compares bytes, like upstream: library.rs:89-99, permutes.rs:3,127-144) for reads and FASTQ text, both strands,
    with and without the single-mismatch level and the position recursion; and the same as the byte-string path alone."""
    rng = random.Random(1000 * L + n_guides + reverse)
    o = 6
    guides = _hybrid_library(rng, L, n_guides, frac)
    n_other = sum(1 for g in guides if any(c not in b"ACGT" for c in g))
    assert 0 < n_other * 2 <= len(guides)
    reads = []
    for _ in range(15000):
        g = bytearray(rng.choice(guides))
        u = rng.random()
        if u < 0.35:
            g[rng.randrange(L)] = rng.choice(b"ACGTNacgtRJ")
        elif u < 0.45:
            g[rng.randrange(L)] = rng.choice(b"ACGTN"); g[rng.randrange(L)] = rng.choice(b"ACGTN")
        elif u < 0.5:
            g = bytearray(rng.choice(b"ACGTN") for _ in range(L))
        pre = bytes(rng.choice(b"ACGTN") if rng.random() < 0.02 else rng.choice(b"ACGT") for _ in range(max(o + rng.choice([0, 0, 0, 1, -1, 2]), 0)))
        tail = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 1, 2, 5, 30])))
        r = pre + bytes(g) + tail
        if rng.random() < 0.03:
            r = r[: rng.randrange(len(r) + 1)]
        reads.append(r)
    if reverse:
        reads = [bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r)) if rng.random() < 0.97 else r for r in reads]
    lib_text, reads_text = _fasta(guides), _reads_fasta(reads)
    lib = _lib(S, lib_text)
    perm = S.Permuter.new(lib.keys())
    info = lib.device(True).info()
    assert info.path == 2 and info.record_bytes == 0 and info.reserved_ == len(guides) - n_other
    off = S.Offset.Reverse(o) if reverse else S.Offset.Forward(o)
    for exact in (False, True):
        for recursion in (True, False):
            want, tot, mat = O.count_text(lib_text, reads_text, reverse, o, exact, recursion)
            for pack in ("device", "fastq", "windows"):
                rs = [r for r in S.parse_fastx(reads_text)]
                ctr = S.Counter.new(iter(rs), lib, None if exact else perm, off, L, recursion, pack=pack, batch=4001)
                assert ctr.guide_counts().tolist() == want, (exact, recursion, pack)
                assert (ctr.total_reads(), ctr.matched_reads()) == (tot, mat)
    # the probing resolver (variant 3: what a library without a core index gets) in a hybrid ctx, many small pushes: the packed pass
    # must leave the pushed reads' offsets alone, the byte-string chain reads them after it (it kept its segment counts on them once:
    # found by tools/fuzz_parity.py)
    want, tot, mat = O.count_text(lib_text, reads_text, reverse, o, False, True)
    for pack, batch in (("device", 7), ("windows", 64)):
        ctr = S.Counter.new(S.parse_fastx(reads_text), lib, perm, off, L, True, pack=pack, batch=batch, options={"variant": 3})
        assert ctr.guide_counts().tolist() == want and (ctr.total_reads(), ctr.matched_reads()) == (tot, mat), (pack, batch)
    # lookups go through the whole library's byte-string tables
    dl = lib.device(True)
    toks = [bytes(g) for g in guides[:50]]
    assert dl.lookup(toks, which=0).tolist() == list(range(50))
    # the byte-string path alone ("hybrid" off) gives the same table
    both = []
    for opts in ({}, {"hybrid": 0}):
        ctr = S.Counter.new(S.parse_fastx(reads_text), lib, perm, off, L, True, pack="device", options=opts)
        both.append(ctr.guide_counts().tolist())
    assert both[0] == both[1]


def test_hybrid_cli_table(tmp_path):
    """the command line with a library of 3000 guides, 30 of them with an 'N': the oracle's table from plain and gzipped FASTQ"""
    import gzip
    import subprocess
    from sgcount_amd import hostlib
    rng = random.Random(8)
    guides = _hybrid_library(rng, 20, 3000, 0.01)
    lib_text = _fasta(guides, b"sg")
    reads = []
    for i in range(30000):
        g = bytearray(rng.choice(guides))
        if rng.random() < 0.3:
            g[rng.randrange(20)] = rng.choice(b"ACGTN")
        reads.append(bytes(rng.choice(b"ACGT") for _ in range(9 + rng.choice([0, 0, 1, -1]))) + bytes(g) + b"GATTACAGATTACA")
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    lp, fq = str(tmp_path / "lib.fa"), str(tmp_path / "r.fastq")
    open(lp, "wb").write(lib_text)
    open(fq, "wb").write(text)
    open(fq + ".gz", "wb").write(gzip.compress(text, 6))
    lib = O.Library(lib_text)
    ctr = O.Counter(lib, O.Permuter(lib), False, 9, 20, True).feed_text(text)
    seen, row = set(), []
    for ident in lib.ids():
        row.append(0 if ident in seen else ctr.get_value(ident)); seen.add(ident)
    want = O.format_results(lib, [row], ["r"], None, False)
    import json
    stats = str(tmp_path / "stats.json")
    for path, extra, scan in ((fq, [], True), (fq, ["--pack", "fastq"], False), (fq + ".gz", [], True), (fq + ".gz", ["--pack", "fastq"], False), (fq, ["--scan-threads", "3", "--scan-block-kb", "8"], True)):
        p = subprocess.run([hostlib.cli_path(), "-l", lp, "-i", path, "-a", "9", "-q", "-n", "r", "--stats-json", stats] + extra, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout.decode() == want, (path, extra)
        assert json.load(open(stats))["samples"][0]["scan_path"] == scan      # plain text and plain gzip: the scanner routes the reads near the 'N' guides itself
    # the other strand, no recursion, exact: the scanner's routing against the GPU's
    rtext = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r)), b"I" * len(r)) for i, r in enumerate(reads))
    open(fq, "wb").write(rtext)
    for flags in (["-r"], ["-r", "-p"], ["-r", "-x"]):
        outs = []
        for extra in ([], ["--pack", "fastq"]):
            p = subprocess.run([hostlib.cli_path(), "-l", lp, "-i", fq, "-a", "14", "-q", "-n", "r", "-z"] + flags + extra, capture_output=True, timeout=300)
            assert p.returncode == 0, p.stderr.decode()
            outs.append(p.stdout.decode())
        ctr = O.Counter(lib, None if "-x" in flags else O.Permuter(lib), True, 14, 20, "-p" not in flags).feed_text(rtext)
        seen, row = set(), []
        for ident in lib.ids():
            row.append(0 if ident in seen else ctr.get_value(ident)); seen.add(ident)
        assert outs[0] == outs[1] == O.format_results(lib, [row], ["r"], None, True), flags
