"""The C++ host side (libsgcount_host.so: FASTX reader, offsetter, gene map, results table, sample names)
against the CPU oracle and the reference's known answers.  No GPU needed: nothing here counts reads."""
import gzip
import os
import random

import numpy as np
import pytest

import _oracle as O
from conftest import DATA


@pytest.fixture(scope="module")
def H():
    from sgcount_amd import hostlib
    hostlib.load()
    return hostlib


def _write(tmp_path, name, data: bytes, gz=False):
    p = os.path.join(str(tmp_path), name)
    with (gzip.open(p, "wb") if gz else open(p, "wb")) as f:
        f.write(data)
    return p


def test_fastx_reader_gz_and_plain(H, tmp_path, example_reads, example_library_text):
    for name, text in example_reads.items():
        want = (text.count(b"\n") // 4, sum(len(l) for l in text.split(b"\n")[1::4]))
        assert H.fastx_stats(os.path.join(DATA, name + ".fastq.gz"))[:2] == want
        assert H.fastx_stats(_write(tmp_path, name + ".fastq", text))[:2] == want
    assert H.library_info(os.path.join(DATA, "library.fasta.gz")) == (100, 20)
    # records far larger than the reader's refill granularity, and a last record without a newline
    big = b">a\n" + b"ACGT" * 3_000_000 + b"\n>b\n" + b"TTGCA" * 1_000_000
    assert H.fastx_stats(_write(tmp_path, "big.fa", big)) == (2, 12_000_000 + 5_000_000, 2)
    assert H.fastx_stats(_write(tmp_path, "big.fa.gz", big, gz=True)) == (2, 17_000_000, 2)


def test_library_errors(H, tmp_path):
    with pytest.raises(H.HostError) as e:                   # library.rs:133-136 duplicates ⇒ panic
        H.library_info(_write(tmp_path, "dup.fa", b">seq.0\nACTG\n>seq.1\nACTG\n"))
    assert e.value.code == 101 and "duplicate" in str(e.value)
    with pytest.raises(H.HostError) as e:                   # library.rs:83
        H.library_info(_write(tmp_path, "bad.fa", b">a\nACTG\n>b\nACT\n"))
    assert e.value.code == 1 and str(e.value) == "Library sequence sizes are inconsistent"
    with pytest.raises(H.HostError) as e:
        H.library_info(_write(tmp_path, "empty.fa", b""))
    assert e.value.code == 101


# ---- offsetter (src/offsetter.rs KATs and oracle agreement) ----------------------------------------
READER = b">seq.0\nACT\n>seq.1\nACC\n>seq.2\nACT\n"
OFFSET_READER = b">seq.0\nAACAAACT\n>seq.1\nAACAAACC\n>seq.2\nAACAAACT\n"
RC_OFFSET_READER = b">seq.0\nAGTTTGTT\n>seq.1\nGGTTTGTT\n>seq.2\nAGTTTGTT\n"


def test_offsetter_kats(H, tmp_path):
    """offsetter.rs:249-328"""
    assert H.minimize_mse(list(np.linspace(0., 10., 11)), list(np.linspace(10., 20., 100))) == (False, 0)
    with pytest.raises(H.HostError) as e:
        H.minimize_mse(list(np.linspace(0., 10., 11)), list(np.linspace(10., 20., 5)))
    assert e.value.code == 1 and str(e.value).startswith("Sequences in reference library are larger")
    lib = _write(tmp_path, "r.fa", READER)
    assert H.entropy_offset_group(lib, [_write(tmp_path, "o.fa", OFFSET_READER)]) == [(False, 5)]
    assert H.entropy_offset_group(lib, [_write(tmp_path, "rc.fa", RC_OFFSET_READER)]) == [(True, 5)]
    assert H.positional_entropy(lib) == O.positional_entropy(READER)


def test_offsetter_examples_match_oracle(H, example_library_text, example_reads):
    lib = os.path.join(DATA, "library.fasta.gz")
    paths = [os.path.join(DATA, n + ".fastq.gz") for n in example_reads]
    got = H.entropy_offset_group(lib, paths)
    assert got == [O.entropy_offset(example_library_text, t) for t in example_reads.values()] == [(False, 5)] * 5
    for n, t in example_reads.items():     # bit-identical f64 entropies (same sequential sums)
        assert H.positional_entropy(os.path.join(DATA, n + ".fastq.gz"), 5000) == O.positional_entropy(t, 5000)


def test_offsetter_random_vs_oracle(H, tmp_path):
    rng = random.Random(11)
    for case in range(12):
        L, pre = rng.choice([8, 12, 20]), rng.randrange(0, 12)
        guides = [bytes(rng.choice(b"ACGT") for _ in range(L)) for _ in range(60)]
        lib_text = b"".join(b">g%d\n%s\n" % (i, g) for i, g in enumerate(guides))
        prefix = bytes(rng.choice(b"ACGT") for _ in range(pre))
        tail = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(5, 25)))
        reads = []
        for i in range(400):
            r = prefix + rng.choice(guides) + tail
            if rng.random() < 0.05:
                r = r[:rng.randrange(1, len(r))]
            if rng.random() < 0.1:
                k = rng.randrange(len(r)); r = r[:k] + b"N" + r[k + 1:]
            if case % 3 == 2:
                r = bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r))
            reads.append(r)
        txt = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
        lp, rp = _write(tmp_path, "l%d.fa" % case, lib_text), _write(tmp_path, "r%d.fq.gz" % case, txt, gz=True)
        for sub in (5000, 37):
            assert H.entropy_offset_group(lp, [rp], sub) == [O.entropy_offset(lib_text, txt, sub)]


def test_offsetter_errors(H, tmp_path):
    lib = _write(tmp_path, "r.fa", b">a\nACGTACGT\n>b\nACGTTTTT\n>c\nAAGTTTTT\n")
    short = _write(tmp_path, "s.fa", b">r\nACG\n>r\nACG\n>r\nACG\n")
    with pytest.raises(H.HostError) as e:                    # offsetter.rs:205 wraps :154-156
        H.entropy_offset_group(lib, [short])
    assert e.value.code == 1 and str(e.value).startswith("Error in entropy offset calculation:\n\nSequences in reference")
    with pytest.raises(H.HostError) as e:                    # offsetter.rs:38 expect("empty reader")
        H.entropy_offset_group(lib, [_write(tmp_path, "e.fa", b"")])
    assert e.value.code == 101
    with pytest.raises(H.HostError) as e:                    # offsetter.rs:195 panic
        H.entropy_offset_group(lib, [os.path.join(str(tmp_path), "nope.fq")])
    assert e.value.code == 101 and "Unable to open file" in str(e.value)


# ---- utils / genemap / results ---------------------------------------------------------------------------
def test_sample_names_match_oracle_and_kats(H):
    kat = ["example/some_name_1.fastq.gz", "example/some_name_2.fastq", "example/some_name_3.fasta.gz",
           "example/some_name_4.fasta", "example/some_name_5.fq.gz", "example/some_name_6.fq",
           "example/some_name_7.fa.gz", "example/some_name_8.fa"]
    assert H.generate_sample_names(kat) == ["some_name_%d" % i for i in range(1, 9)]             # utils.rs:55-81
    dup = ["example/some_name_1.fastq.gz", "example/some_name_1.fastq"] + kat[2:]
    assert H.generate_sample_names(dup) == ["Sample.%d" % i for i in range(8)]                   # utils.rs:84-104
    odd = ["a.fq.gz.gz", "/x/y/b.fasta.fa", "c.fastq.fq", "noext", "d.gz.fq", "dir.fq/e.fa", "f.fa.fa.gz"]
    assert H.generate_sample_names(odd) == O.generate_sample_names(odd)[0] == ["a", "b.fasta", "c.fastq", "noext", "d.gz", "e", "f"]


def test_genemap_matches_oracle_and_kats(H, tmp_path):
    text = b"gene1\tsgrna1\ngene2\tsgrna2\ngene3\tsgrna3\n"
    for k in (b"sgrna1", b"sgrna3", b"sgrna9"):
        assert H.genemap_get(k, text=text) == O.GeneMap(text).get(k)
    g2s = os.path.join(DATA, "g2s.txt")
    assert H.genemap_get(b"lib.0", path=g2s) == b"gene.0" and H.genemap_get(b"lib.99", path=g2s) == b"gene.9"   # genemap.rs:151-156
    ok = _write(tmp_path, "ok.fa", b">sgrna1\nACTG\n>sgrna2\ngtca\n>sgrna3\nTCAG\n")
    bad = _write(tmp_path, "bad.fa", b">sgrna1\nACTG\n>sgrna4\ngtca\n")
    assert H.genemap_missing(text, ok) is None and H.genemap_missing(text, bad) == b"sgrna4"     # genemap.rs:133-148
    with pytest.raises(H.HostError) as e:
        H.genemap_get(b"x", text=b"gene1 sgrna1\n")
    assert e.value.code == 101 and "Missing" in str(e.value)
    with pytest.raises(H.HostError) as e:
        H.genemap_get(b"x", text=b"g\ts\nh\ts\n")
    assert e.value.code == 101 and str(e.value) == "Duplicate sgRNA key found in gene map: s"
    with pytest.raises(H.HostError) as e:
        H.genemap_get(b"x", path=os.path.join(str(tmp_path), "missing.txt"))
    assert e.value.code == 1 and str(e.value).startswith("Provided gene mapping path doesn't exist")
    assert H.genemap_get(b"s", text=b"g\ts\r\n") == b"g"


def test_results_match_oracle(H, tmp_path):
    assert H.generate_columns(["A", "B"]) == "Guide\tA\tB"                                          # results.rs:134-139
    assert H.generate_columns(["A", "B"], True) == "Guide\tGene\tA\tB" == O.generate_columns(["A", "B"], True)
    rng = random.Random(3)
    lib_text = gzip.open(os.path.join(DATA, "library.fasta.gz")).read()
    lib_text += b">lib.0\n" + b"A" * 20 + b"\n"            # a duplicated id: both rows print the pooled count
    lp = _write(tmp_path, "lib.fa", lib_text)
    olib = O.Library(lib_text)
    gm_text = open(os.path.join(DATA, "g2s.txt"), "rb").read()
    for n_samples in (1, 3):
        counts = [[rng.choice([0, 0, 1, 7, 123456789012]) for _ in range(101)] for _ in range(n_samples)]
        names = ["s%d" % i for i in range(n_samples)]
        for gm in (None, gm_text):
            for z in (False, True):
                want = O.format_results(olib, counts, names, O.GeneMap(gm) if gm else None, z)
                assert H.format_results(lp, counts, names, gm, z) == want
    with pytest.raises(H.HostError) as e:                                                            # results.rs:59
        H.format_results(lp, [[1] * 101], ["s"], b"gene.0\tlib.0\n", True)
    assert e.value.code == 101


def test_text_feeder_delivers_every_byte_once(tmp_path):
    """The byte source of the count path's text mode (multi-threaded pread / one inflating producer, slices cut at
    their last newline, the rest carried): every byte is delivered exactly once and in order, the per-slice newline
    counts are right, whatever the slice size and thread count — plain and .gz, with and without a final newline,
    with lines longer than a slice."""
    import gzip
    import random
    from sgcount_amd import hostlib
    rng = random.Random(31)
    recs = []
    for i in range(3000):
        n = rng.choice([0, 1, 20, 90, 150, 400])
        seq = bytes(rng.choice(b"ACGTN") for _ in range(n))
        qual = bytes(rng.choice(b"@+I#5") for _ in range(n))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, seq, qual))
    recs.append(b"@long\n" + b"A" * 200_000 + b"\n+\n" + b"I" * 200_000 + b"\n")       # lines longer than a 64 KiB slice
    text = b"".join(recs)

    def fnv(b):
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    for body in (text, text[:-1]):
        want_lines = body.count(b"\n") + (0 if body.endswith(b"\n") else 1)
        want = (len(body), want_lines, fnv(body))
        plain = tmp_path / "t.fastq"
        plain.write_bytes(body)
        gz = tmp_path / "t.fastq.gz"
        with gzip.open(gz, "wb", compresslevel=1) as f:
            f.write(body)
        for path, is_gz in ((plain, False), (gz, True)):
            for slice_bytes, threads in ((1 << 16, 1), (1 << 16, 3), (100_000, 4), (1 << 22, 2)):
                parts, nbytes, lines, h, first, gzflag = hostlib.text_feeder_walk(str(path), slice_bytes, threads)
                assert (nbytes, lines, h) == want and first == ord("@") and gzflag == is_gz
                assert parts >= 1
    empty = tmp_path / "empty.fastq"
    empty.write_bytes(b"")
    assert hostlib.text_feeder_walk(str(empty))[:3] == (0, 0, 0)


def test_text_feeder_inflates_bgzf_members_in_parallel(tmp_path):
    """A BGZF file goes through several inflating threads (members are independent deflate streams placed by their ISIZE
    trailers): same bytes, same order, right newline counts; members of uneven size, no end-of-file marker, a corrupt member."""
    import gzip
    import random
    from sgcount_amd import hostlib
    from sgcount_amd.bgzf import bgzf_bytes
    rng = random.Random(5)
    recs = []
    for i in range(6000):
        n = rng.choice([0, 1, 20, 90, 150, 400])
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGTN") for _ in range(n)), bytes(rng.choice(b"@+I#5") for _ in range(n))))
    text = b"".join(recs)

    def fnv(b):
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    for body, block, marker in ((text, 65280, True), (text[:-1], 4000, False), (text, 65536, True), (text[:70000], 100, True)):
        want = (len(body), body.count(b"\n") + (0 if body.endswith(b"\n") else 1), fnv(body))
        p = tmp_path / "t.fastq.gz"
        blob = bgzf_bytes(body, block, eof_marker=marker)
        p.write_bytes(blob)
        assert gzip.decompress(blob) == body                     # it is ordinary multi-member gzip too
        for slice_bytes, threads in ((1 << 16, 1), (1 << 16, 4), (200_000, 3), (1 << 22, 8)):
            parts, nbytes, lines, h, first, flag = hostlib.text_feeder_walk(str(p), slice_bytes, threads)
            assert (nbytes, lines, h) == want and first == ord("@") and flag == "bgzf", (block, slice_bytes, threads)
    bad = bytearray(bgzf_bytes(text, 30000))
    bad[len(bad) // 2] ^= 0x55
    (tmp_path / "bad.fastq.gz").write_bytes(bytes(bad))
    with pytest.raises(hostlib.HostError):
        hostlib.text_feeder_walk(str(tmp_path / "bad.fastq.gz"), 1 << 16, 4)


def _pack_host(seqs, L, reverse, o, recursion):
    """sgc_pack_reads_host (the scalar packer of the C ABI) on a list of reads -> uint64 array [n, words]"""
    import ctypes as C
    import numpy as np
    from sgcount_amd import _ffi
    lib = _ffi.load()
    flat = b"".join(seqs)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in seqs], out=offs[1:])
    words = 2 if L > 23 else 1
    out = np.zeros((len(seqs), words), dtype=np.uint64)
    buf = (C.c_uint8 * max(len(flat), 1)).from_buffer_copy(flat or b"\0")
    _ffi.check(lib.sgc_pack_reads_host(C.cast(buf, C.c_void_p), offs.ctypes.data, len(seqs), L, int(reverse), o, int(recursion),
                                       out.ctypes.data))
    return out


def test_scanner_records_equal_the_host_packer(tmp_path):
    """The scan path's record source (FastqScanner: memory-mapped text, blocks taken by several threads, line numbers chained
    across blocks, SIMD fast path + sgc_pack_one) gives, for every read and in order, exactly the record of the scalar host
    packer — both strands, recursion on/off, one- and two-word records, offsets 0/1/30, 'N' and lowercase bytes, short and
    empty reads, lines that straddle blocks, CRLF terminators, no final newline, trailing blank lines."""
    import random
    import numpy as np
    from sgcount_amd import hostlib
    rng = random.Random(77)
    seqs = []
    for i in range(5000):
        n = rng.choice([0, 1, 19, 20, 21, 22, 23, 40, 51, 52, 53, 60, 90, 150, 150, 150, 400])
        alphabet = b"ACGT" if rng.random() < 0.8 else rng.choice([b"ACGTN", b"ACGTNacgtRY", b"ACGTJ"])
        seqs.append(bytes(rng.choice(alphabet) for _ in range(n)))
    seqs.append(b"ACGT" * 40_000)                                   # a line much longer than a block
    def fastq(eol, final_newline=True, blank_tail=0):
        out = []
        for i, s in enumerate(seqs):
            q = bytes(rng.choice(b"@+I#5") for _ in range(len(s)))
            out.append(b"@r%d" % i + eol + s + eol + b"+" + eol + q + eol)
        t = b"".join(out)
        if not final_newline:
            t = t[: -len(eol)]
        return t + b"\n" * blank_tail
    p = tmp_path / "r.fastq"
    for eol, final_nl, tail in ((b"\n", True, 0), (b"\n", False, 0), (b"\r\n", True, 0), (b"\n", True, 3)):
        p.write_bytes(fastq(eol, final_nl, tail))
        for L, rev, o, rec in ((20, False, 30, True), (20, True, 30, True), (20, False, 0, True), (20, True, 1, False),
                               (23, False, 5, True), (24, True, 7, True), (30, False, 30, False), (4, False, 2, True)):
            want = _pack_host(seqs, L, rev, o, rec)
            for threads, block, source in ((1, 1 << 22, 0), (3, 4096, 1), (4, 1 << 16, 2), (2, 4096, 2)):
                got, lines = hostlib.scan_records(str(p), L, rev, o, rec, threads=threads, block_bytes=block, source=source)
                assert lines == 4 * len(seqs)
                assert got.shape == want.shape and np.array_equal(got, want), (eol, final_nl, tail, L, rev, o, rec, threads, block, source)
    # the scanner declines what is not plain FASTQ text; malformed and truncated records panic (exit code 101)
    (tmp_path / "lib.fa").write_bytes(b">a\nACGT\n")
    assert hostlib.scan_records(str(tmp_path / "lib.fa"), 4) is None
    (tmp_path / "empty.fastq").write_bytes(b"")
    assert hostlib.scan_records(str(tmp_path / "empty.fastq"), 4) is None
    good = fastq(b"\n")
    lines = good.split(b"\n")
    bad = list(lines)
    bad[4 * 700 + 2] = b"-"
    (tmp_path / "bad.fastq").write_bytes(b"\n".join(bad))
    with pytest.raises(hostlib.HostError) as e:
        hostlib.scan_records(str(tmp_path / "bad.fastq"), 20, threads=3, block_bytes=4096)
    assert e.value.code == 101 and "line %d " % (4 * 700 + 3) in str(e.value)
    (tmp_path / "trunc.fastq").write_bytes(b"\n".join(lines[: 4 * 900 + 2]) + b"\n")
    with pytest.raises(hostlib.HostError) as e:
        hostlib.scan_records(str(tmp_path / "trunc.fastq"), 20, threads=2, block_bytes=8192)
    assert e.value.code == 101 and "truncated" in str(e.value)


def test_scanner_packs_a_gzip_stream_like_its_text(tmp_path):
    """FastqScanner on ONE gzip stream (not BGZF): the chunks of the compressed file are decoded speculatively, stitched in order,
    resolved into the worker's own buffer and packed there, with the unfinished line handed from chunk to chunk — the records and
    the line count must be those of the plain text, for every chunking (chunks much smaller than a deflate block: decoded in
    order; several members; no final newline; trailing blank lines; CRLF), and a damaged stream must be reported."""
    import gzip
    import numpy as np
    from sgcount_amd import hostlib
    rng = random.Random(99)
    seqs = []
    for i in range(30000):
        n = rng.choice([0, 1, 19, 20, 21, 22, 23, 40, 51, 52, 53, 60, 90, 150, 150, 150, 400])
        alphabet = b"ACGT" if rng.random() < 0.8 else rng.choice([b"ACGTN", b"ACGTNacgtRY", b"ACGTJ"])
        seqs.append(bytes(rng.choice(alphabet) for _ in range(n)))
    seqs.append(b"ACGT" * 40_000)                                   # a line much longer than a chunk's text
    def fastq(eol, final_newline=True, blank_tail=0):
        out = []
        for i, s in enumerate(seqs):
            q = bytes(rng.choice(b"@+I#5") for _ in range(len(s)))
            out.append(b"@r%d" % i + eol + s + eol + b"+" + eol + q + eol)
        t = b"".join(out)
        if not final_newline:
            t = t[: -len(eol)]
        return t + b"\n" * blank_tail
    p = tmp_path / "r.fastq.gz"
    for eol, final_nl, tail, members, level in ((b"\n", True, 0, 1, 6), (b"\n", False, 0, 3, 1), (b"\r\n", True, 0, 1, 9), (b"\n", True, 3, 2, 6)):
        text = fastq(eol, final_nl, tail)
        cuts = sorted(rng.randrange(len(text)) for _ in range(members - 1))
        parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
        p.write_bytes(b"".join(gzip.compress(x, compresslevel=level) for x in parts))
        for L, rev, o, rec in ((20, False, 30, True), (20, True, 30, True), (24, True, 7, True), (4, False, 2, True)):
            want = _pack_host(seqs, L, rev, o, rec)
            for threads, block in ((2, 1 << 22), (3, 4096), (5, 1 << 16), (4, 700)):
                got, lines = hostlib.scan_records(str(p), L, rev, o, rec, threads=threads, block_bytes=block, cap=len(seqs) + 16)
                assert lines == 4 * len(seqs), (eol, final_nl, tail, members, threads, block)
                assert got.shape == want.shape and np.array_equal(got, want), (eol, final_nl, tail, members, L, rev, o, rec, threads, block)
    # BGZF (every member announces its size): runs of whole members, inflated by zlib in the workers; BGZF followed by a plain gzip
    # member is not BGZF throughout and takes the speculative decoder; a damaged member is reported
    from sgcount_amd.bgzf import bgzf_bytes
    text = fastq(b"\n")
    want = _pack_host(seqs, 20, False, 30, True)
    for name, blob in (("b1.gz", bgzf_bytes(text, 65280)), ("b2.gz", bgzf_bytes(text, 12000, eof_marker=False)),
                       ("b3.gz", bgzf_bytes(text[: len(text) // 2], 20000, eof_marker=False) + gzip.compress(text[len(text) // 2:]))):
        (tmp_path / name).write_bytes(blob)
        for threads, block in ((2, 1 << 22), (4, 1 << 16), (3, 5000)):
            got, lines = hostlib.scan_records(str(tmp_path / name), 20, False, 30, True, threads=threads, block_bytes=block, cap=len(seqs) + 16)
            assert lines == 4 * len(seqs) and np.array_equal(got, want), (name, threads, block)
    # blank lines at the end that sit in members (chunks) of their own, behind a CRLF text and behind an LF text: not records; the same
    # blank lines with a record behind them: a malformed header
    from sgcount_amd.bgzf import member, EOF_MARKER
    for eol in (b"\n", b"\r\n"):
        body = fastq(eol)
        want_e = _pack_host(seqs, 20, False, 30, True)
        (tmp_path / "tail.gz").write_bytes(bgzf_bytes(body, 20000, eof_marker=False) + member(b"\n") + member(b"\n\n") + EOF_MARKER)
        for threads, block in ((3, 512), (2, 1 << 22)):
            got, lines = hostlib.scan_records(str(tmp_path / "tail.gz"), 20, False, 30, True, threads=threads, block_bytes=block, cap=len(seqs) + 16)
            assert lines == 4 * len(seqs) and np.array_equal(got, want_e), (eol, threads, block)
        (tmp_path / "mid.gz").write_bytes(bgzf_bytes(body, 20000, eof_marker=False) + member(b"\n") + member(b"\n") + bgzf_bytes(b"@x" + eol + b"ACGT" + eol + b"+" + eol + b"IIII" + eol, 20000))
        with pytest.raises(hostlib.HostError) as e:
            hostlib.scan_records(str(tmp_path / "mid.gz"), 20, False, 30, True, threads=3, block_bytes=512, cap=len(seqs) + 16)
        assert e.value.code == 101 and "line %d " % (4 * len(seqs) + 1) in str(e.value)
    blob = bytearray(bgzf_bytes(text, 30000)); blob[len(blob) // 2] ^= 0x10
    (tmp_path / "bbad.gz").write_bytes(bytes(blob))
    with pytest.raises(hostlib.HostError):
        hostlib.scan_records(str(tmp_path / "bbad.gz"), 20, False, 30, True, threads=3, block_bytes=1 << 16, cap=len(seqs) + 16)
    # not FASTQ inside: declined (the record reader decides); one thread: declined (the sequential inflater serves it)
    (tmp_path / "lib.fa.gz").write_bytes(gzip.compress(b">a\nACGT\n" * 50))
    assert hostlib.scan_records(str(tmp_path / "lib.fa.gz"), 4, threads=3) is None
    assert hostlib.scan_records(str(p), 20, threads=1, cap=len(seqs) + 16) is None
    # a damaged stream, a truncated stream, a malformed record and a truncated record inside a good stream
    good = gzip.compress(fastq(b"\n"), compresslevel=6)
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x55
    (tmp_path / "bad.gz").write_bytes(bytes(bad))
    with pytest.raises(hostlib.HostError):
        hostlib.scan_records(str(tmp_path / "bad.gz"), 20, threads=3, block_bytes=1 << 16, cap=len(seqs) + 16)
    (tmp_path / "trunc.gz").write_bytes(good[: len(good) * 2 // 3])
    with pytest.raises(hostlib.HostError):
        hostlib.scan_records(str(tmp_path / "trunc.gz"), 20, threads=3, block_bytes=1 << 16, cap=len(seqs) + 16)
    # a damaged FILE is never blamed on the sample: text that breaks the 4-line cycle is reported as a malformed record (code 101)
    # only once the member's CRC has vouched for it
    for seed in range(24):
        r2 = random.Random(seed)
        bad = bytearray(good)
        for _ in range(r2.choice([1, 1, 3])):
            bad[r2.randrange(20, len(bad) - 8)] ^= 1 << r2.randrange(8)
        (tmp_path / "flip.gz").write_bytes(bytes(bad))
        with pytest.raises(hostlib.HostError) as e:
            hostlib.scan_records(str(tmp_path / "flip.gz"), 20, threads=3, block_bytes=1 << 15, cap=len(seqs) + 16)
        assert e.value.code == 1, (seed, str(e.value))
    lines = fastq(b"\n").split(b"\n")
    mal = list(lines); mal[4 * 700 + 2] = b"-"
    (tmp_path / "mal.gz").write_bytes(gzip.compress(b"\n".join(mal)))
    with pytest.raises(hostlib.HostError) as e:
        hostlib.scan_records(str(tmp_path / "mal.gz"), 20, threads=3, block_bytes=8192, cap=len(seqs) + 16)
    assert e.value.code == 101 and "line %d " % (4 * 700 + 3) in str(e.value)
    (tmp_path / "short.gz").write_bytes(gzip.compress(b"\n".join(lines[: 4 * 900 + 2]) + b"\n"))
    with pytest.raises(hostlib.HostError) as e:
        hostlib.scan_records(str(tmp_path / "short.gz"), 20, threads=2, block_bytes=8192, cap=len(seqs) + 16)
    assert e.value.code == 101 and "truncated" in str(e.value)


def test_bgzf_followed_by_plain_gzip_members_is_read_as_gzip(tmp_path):
    """bgzip output concatenated with ordinary gzip output is legal multi-member gzip (zlib and the reference's flate2 read
    it): the member chain is checked when the file is opened and such a file takes the sequential inflater instead of
    aborting with "corrupt BGZF member header" half way."""
    import gzip
    from sgcount_amd import hostlib
    from sgcount_amd.bgzf import bgzf_bytes
    a = b"".join(b"@a%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(4000))
    b = b"".join(b"@b%d\nTTTTGGGGCC\n+\nIIIIIIIIII\n" % i for i in range(3000))
    p = tmp_path / "mixed.fastq.gz"
    p.write_bytes(bgzf_bytes(a, 20000, eof_marker=False) + gzip.compress(b))
    body = a + b

    def fnv(x):
        h = 1469598103934665603
        for c in x:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    parts, nbytes, lines, h, first, flag = hostlib.text_feeder_walk(str(p), 1 << 16, 4)
    assert (nbytes, lines, h) == (len(body), body.count(b"\n"), fnv(body)) and first == ord("@") and flag is True


def test_plain_gzip_streams_are_inflated_by_several_threads(tmp_path):
    """A plain .gz (ONE deflate stream — what sequencers and `gzip` write, and what the reference inflates on the sample's single
    thread, count.rs:24) is cut into chunks of compressed bytes that are decoded speculatively in parallel and stitched in order
    (sgh_inflate.cpp + TextFeeder::run_pgz): every byte exactly once and in order, whatever the compression level, the chunk size
    (smaller than a deflate block: blocks span chunks; larger than the file: one chunk), the slice size and the thread count;
    several gzip members; a member boundary inside a chunk; data that is not text (every chunk then falls back to in-order
    decoding); and a corrupt or truncated stream is an error, never a silent miscount."""
    import gzip
    import random
    from sgcount_amd import hostlib
    rng = random.Random(17)
    recs = []
    for i in range(12000):
        n = rng.choice([36, 75, 100, 150, 151])
        recs.append(b"@r%d extra words\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGTN") for _ in range(n)), bytes(rng.choice(b"FFFFF:,#I") for _ in range(n))))
    text = b"".join(recs)

    def fnv(b):
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    def check(blob, body, combos, expect_pgz=True):
        p = tmp_path / "t.fastq.gz"
        p.write_bytes(blob)
        want = (len(body), body.count(b"\n") + (0 if body.endswith(b"\n") or not body else 1), fnv(body))
        for slice_bytes, threads, chunk in combos:
            info = {}
            parts, nbytes, lines, h, first, flag = hostlib.text_feeder_walk(str(p), slice_bytes, threads, chunk, info)
            assert (nbytes, lines, h) == want and flag is True and info["pgz"] == expect_pgz, (slice_bytes, threads, chunk, info)
        return info
    combos = ((1 << 16, 4, 0), (1 << 16, 3, 20_000), (100_000, 8, 65_536), (1 << 22, 2, 300_000), (1 << 18, 5, 7_000))
    for level in (1, 6, 9):
        check(gzip.compress(text, level), text, combos)
    # several members (python's gzip, concatenated; zlib reads them as one stream), one of them tiny, one empty
    third = len(text) // 3
    multi = gzip.compress(text[:third], 6) + gzip.compress(b"", 6) + gzip.compress(text[third:third + 10], 1) + gzip.compress(text[third + 10:], 9)
    check(multi, text, combos)
    # no final newline
    check(gzip.compress(text[:-1], 6), text[:-1], combos[:3])
    # very repetitive text (the bench's synthetic FASTQ: constant adapters, qualities all 'I' — long overlapping copies at
    # distance 1 and markers that live long)
    from sgcount_amd import synth
    rep = synth.fastq_host(synth.library(500, 20), 0, 30000)
    for level in (1, 9):
        check(gzip.compress(rep, level), rep, ((1 << 16, 4, 9_000), (1 << 20, 6, 0), (1 << 18, 3, 50_000)))
    # bytes that are not text: no speculative start is ever accepted, every chunk is decoded in order — still every byte, once
    blob = bytes(rng.randrange(256) for _ in range(200_000)) + text[:300_000]
    info = check(gzip.compress(b"@" + blob, 6), b"@" + blob, ((1 << 16, 4, 30_000),))
    assert info["fallbacks"] >= 3
    # a file too small for the parallel decoder goes through zlib
    check(gzip.compress(b"@r\nAC\n+\nII\n", 6), b"@r\nAC\n+\nII\n", ((1 << 16, 4, 0),), expect_pgz=False)
    # corrupt and truncated streams
    good = gzip.compress(text, 6)
    for bad in (good[: len(good) // 2], good[:-4], good[: len(good) // 3] + bytes([good[len(good) // 3] ^ 0x10]) + good[len(good) // 3 + 1:]):
        (tmp_path / "bad.fastq.gz").write_bytes(bad)
        with pytest.raises(hostlib.HostError):
            hostlib.text_feeder_walk(str(tmp_path / "bad.fastq.gz"), 1 << 16, 4, 40_000)


def test_folded_crc32_equals_zlib():
    """crc32_fast (carry-less-multiply folding; every inflated byte of the gzip / BGZF readers goes through it) == zlib.crc32
    for every length around the 16- and 64-byte boundaries of the folding and any starting value."""
    import ctypes as C
    import random
    import zlib
    from sgcount_amd import hostlib
    L = hostlib.load()
    L.sgh_crc32.restype = C.c_uint32
    rng = random.Random(2)
    for n in list(range(0, 200)) + [255, 256, 257, 1000, 4095, 4096, 65537, (1 << 20) + 3]:
        b = bytes(rng.randrange(256) for _ in range(min(n, 300))) * (n // 300 + 1)
        b = b[:n]
        for init in (0, 0xDEADBEEF, 0xFFFFFFFF):
            assert L.sgh_crc32(C.c_uint32(init), b, C.c_uint64(n)) == zlib.crc32(b, init), (n, init)


# ---- the command line's argument forms (clap 4 derive over src/main.rs:54-108) — parsed only: `--dry-run-args` prints what the line
# parsed to and exits before any file or device is touched
def _dry(argv):
    import json
    import subprocess
    from sgcount_amd import hostlib as HL
    r = subprocess.run([HL.cli_path(), "--dry-run-args"] + argv, capture_output=True, text=True, timeout=60)
    return r.returncode, (json.loads(r.stdout) if r.returncode == 0 else None), r.stderr


CLI_ACCEPTED = [
    # (argv, expected fields)
    (["-l", "lib.fa", "-i", "a.fq", "b.fq", "--offset=5", "-xzq"],
     {"library_path": "lib.fa", "input_paths": ["a.fq", "b.fq"], "offset": 5, "exact": True, "include_zero": True, "quiet": True}),
    (["-l", "lib.fa", "-i", "a.fq", "-a5"], {"offset": 5, "exact": False}),
    (["-l", "lib.fa", "-i", "a.fq", "-a=5"], {"offset": 5}),
    (["-l", "lib.fa", "-i", "a.fq", "-xa", "5"], {"offset": 5, "exact": True}),
    (["-l", "lib.fa", "-i", "a.fq", "-xa5", "-n", "s1"], {"offset": 5, "exact": True, "sample_names": ["s1"]}),
    (["-llib.fa", "-ia.fq", "b.fq", "-t=4", "-prz"], {"library_path": "lib.fa", "input_paths": ["a.fq", "b.fq"], "threads": 4,
                                                      "no_position_recursion": True, "reverse": True, "include_zero": True}),
    (["--library-path=lib.fa", "--input-paths=a.fq", "b.fq", "-i", "c.fq"], {"input_paths": ["a.fq", "b.fq", "c.fq"]}),
    (["--library-path", "lib.fa", "--input-paths", "a.fq", "--sample-names", "x", "--output-path=o.tsv", "--genemap", "g.txt",
      "--subsample=100", "--threads", "2", "--no-position-recursion", "--reverse", "--exact", "--quiet", "--include-zero"],
     {"sample_names": ["x"], "output_path": "o.tsv", "genemap": "g.txt", "subsample": 100, "threads": 2, "no_position_recursion": True,
      "reverse": True, "exact": True, "quiet": True, "include_zero": True}),
    (["-l", "lib.fa", "-i", "a.fq", "--"], {"input_paths": ["a.fq"]}),
    (["-l", "lib.fa", "-i", "-", "-o", "-"], {"input_paths": ["-"], "output_path": "-"}),          # a lone dash is a value
    (["-i", "a.fq", "-l", "lib.fa", "-n", "s", "-n", "t", "-i", "b.fq"], {"input_paths": ["a.fq", "b.fq"], "sample_names": ["s", "t"]}),
    (["-l", "lib.fa", "-i", "a.fq", "--include-permutations"], {"exact": False}),                  # BASELINE.json's name for the default
]
CLI_REJECTED = [
    # (argv, a piece of clap's message) — all exit with code 2
    (["-l", "lib.fa", "-i", "a.fq", "--exact=1"], "unexpected value '1' for '--exact' found; no more were expected"),
    (["-l", "lib.fa", "-i", "a.fq", "-x=1"], "unexpected value '1' for '--exact' found"),
    (["-l", "lib.fa", "-i", "a.fq", "-x", "-x"], "the argument '--exact' cannot be used multiple times"),
    (["-l", "lib.fa", "-l", "lib2.fa", "-i", "a"], "the argument '--library-path <LIBRARY_PATH>' cannot be used multiple times"),
    (["-l", "lib.fa", "-i", "a.fq", "--", "extra"], "unexpected argument 'extra' found"),
    (["-l", "lib.fa", "-i", "a.fq", "-a"], "a value is required for '--offset <OFFSET>' but none was supplied"),
    (["-l", "lib.fa", "-i", "a.fq", "-a", "-x"], "a value is required for '--offset <OFFSET>' but none was supplied"),
    (["-l", "lib.fa", "-i", "a.fq", "-a", "-5"], "unexpected argument '-5' found"),
    (["-l", "lib.fa", "-i", "a.fq", "--bogus"], "unexpected argument '--bogus' found"),
    (["-l", "lib.fa", "-i", "a.fq", "-xk"], "unexpected argument '-k' found"),
    (["-l", "lib.fa", "-x", "-i"], "a value is required for '--input-paths <INPUT_PATHS>...' but none was supplied"),
    (["-l", "lib.fa", "-i", "a.fq", "-a", "5x"], "invalid value '5x' for '--offset <OFFSET>': invalid digit found in string"),
    (["-l", "lib.fa", "-i", "a.fq", "-t", ""], "invalid value '' for '--threads <THREADS>': cannot parse integer from empty string"),
    (["stray", "-l", "lib.fa", "-i", "a"], "unexpected argument 'stray' found"),
    (["-i", "a.fq"], "the following required arguments were not provided"),
    (["-l", "lib.fa"], "--input-paths <INPUT_PATHS>..."),
]


@pytest.mark.parametrize("argv,want", CLI_ACCEPTED, ids=[" ".join(a) for a, _ in CLI_ACCEPTED])
def test_cli_argument_forms_accepted(argv, want):
    rc, got, err = _dry(argv)
    assert rc == 0, err
    for k, v in want.items():
        assert got[k] == v, (k, got)


@pytest.mark.parametrize("argv,msg", CLI_REJECTED, ids=[" ".join(a) for a, _ in CLI_REJECTED])
def test_cli_argument_forms_rejected(argv, msg):
    rc, _, err = _dry(argv)
    assert rc == 2, (rc, err)
    assert msg in err, err


def test_cli_help_and_version_win():
    import subprocess
    from sgcount_amd import hostlib as HL
    for argv in (["--help"], ["-l", "x", "-h", "--bogus-after-help-is-not-reached"][:3], ["-V"], ["-xV"]):
        r = subprocess.run([HL.cli_path()] + argv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout, (argv, r.stderr)


# ---- reader decision #3: a FASTQ stream that ends behind a separator line ends with a record whose quality line is empty
LAST_RECORD_TAILS = [
    # (what follows 50 whole records, records expected (None = an error), why)
    (b"@e\n\n+\n\n", 51), (b"@e\n\n+\n", 51), (b"@e\n\n+", 51), (b"@e\n\n+\n\n\n\n", 51), (b"@e\r\n\r\n+\r\n", 51), (b"@e\r\n\r\n+\r\n\r\n", 51),
    (b"@e\nACGTACGTAC\n+\n", 51), (b"@e\nACGTACGTAC\n+", 51),
    (b"\n\n\n", 50), (b"", 50),
    (b"@e\n\n", None), (b"@e\n", None), (b"@e", None), (b"@e\nACGT\n", None),
]


@pytest.mark.parametrize("tail,want", LAST_RECORD_TAILS, ids=[repr(t) for t, _ in LAST_RECORD_TAILS])
def test_last_record_with_an_empty_quality_line(tmp_path, tail, want):
    """Every reader takes the same decision on how a FASTQ stream may end: the oracle (fx_next), the Python mirror (parse_fastx), the
    record reader (FastxReader), and the scanner on plain text, one gzip stream and BGZF.  (fxread's own behaviour is unpinned:
    SURVEY §8c; the GPU text parser takes the same decision in tests/test_ingest_gpu.py.)"""
    import gzip
    from sgcount_amd import hostlib, host as S, bgzf
    text = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(50)) + tail
    lib = b">g\nACGTACGTAC\n"
    paths = {"plain": tmp_path / "s.fastq", "gz": tmp_path / "s.fastq.gz", "bgzf": tmp_path / "b.fastq.gz"}
    paths["plain"].write_bytes(text)
    paths["gz"].write_bytes(gzip.compress(text))
    paths["bgzf"].write_bytes(bgzf.bgzf_bytes(text, block=300))
    if want is None:
        with pytest.raises(Exception):
            O.count_text(lib, text, False, 0, True, True)
        with pytest.raises(ValueError):
            list(S.parse_fastx(text))
        for p in paths.values():
            with pytest.raises(hostlib.HostError) as e:
                hostlib.fastx_stats(str(p))
            assert e.value.code == 101
            with pytest.raises(hostlib.HostError) as e:
                hostlib.scan_records(str(p), 10, threads=2, block_bytes=4096)
            assert e.value.code == 101
        return
    _, total, matched = O.count_text(lib, text, False, 0, True, True)
    assert total == want and matched == 50 + (1 if b"ACGTACGTAC" in tail else 0)
    assert len(list(S.parse_fastx(text))) == want
    for kind, p in paths.items():
        assert hostlib.fastx_stats(str(p))[0] == want, kind
        got, lines = hostlib.scan_records(str(p), 10, threads=2, block_bytes=4096)
        assert len(got) == want, kind
