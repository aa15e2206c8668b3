"""End-to-end: the `sgcount-hip` command line (C++ host over the C ABI, GPU count path) against tables built by
the CPU oracle.  Row order is library-file order on both sides, so the comparison is byte for byte."""
import gzip
import os
import subprocess

import pytest

import _oracle as O
from conftest import DATA

pytestmark = pytest.mark.gpu

NAMES = ["sequence", "zero.sequence", "diff.sequence", "offset", "offset_clipped"]
LIB = os.path.join(DATA, "library.fasta.gz")


@pytest.fixture(scope="module")
def cli():
    from sgcount_amd import hostlib
    return hostlib.cli_path()


def run(cli, *args):
    p = subprocess.run([cli, *args], capture_output=True, timeout=300)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def oracle_table(lib_text, reads_texts, names, offsets, exact, recursion, genemap_text=None, include_zero=False):
    lib = O.Library(lib_text)
    cols = []
    for text, (rev, idx) in zip(reads_texts, offsets):
        c, _, _ = O.count_text(lib_text, text, rev, idx, exact, recursion)
        per_guide_pooled = c       # count_text already reports get_value(id) per guide (pooled)
        cols.append(per_guide_pooled)
    # format_results pools by id itself, so hand it un-pooled per-guide counts: rebuild them from one Counter
    cols = []
    for text, (rev, idx) in zip(reads_texts, offsets):
        perm = None if exact else O.Permuter(lib)
        ctr = O.Counter(lib, perm, rev, idx, lib.size(), recursion).feed_text(text)
        seen, row = set(), []
        for ident in lib.ids():
            row.append(0 if ident in seen else ctr.get_value(ident))
            seen.add(ident)
        cols.append(row)
    return O.format_results(lib, cols, names, O.GeneMap(genemap_text) if genemap_text else None, include_zero)


def test_single_sample_fixed_offset_exact(cli, example_library_text, example_reads):
    rc, out, err = run(cli, "-l", LIB, "-i", os.path.join(DATA, "sequence.fastq.gz"), "-a", "5", "-x", "-q")
    assert rc == 0, err
    assert out == oracle_table(example_library_text, [example_reads["sequence"]], ["sequence"], [(False, 5)], True, True)
    assert err == ""


@pytest.mark.parametrize("pack", ["scan", "fastq", "device", "host"])
def test_all_samples_auto_offset_genemap_zero(cli, pack, tmp_path, example_library_text, example_reads):
    paths = [os.path.join(DATA, n + ".fastq.gz") for n in NAMES]
    g2s = os.path.join(DATA, "g2s.txt")
    outp = os.path.join(str(tmp_path), "out.tsv")
    rc, out, err = run(cli, "-l", LIB, "-i", *paths, "-g", g2s, "-z", "-t", "3", "-o", outp, "--pack", pack)
    assert rc == 0, err
    want = oracle_table(example_library_text, [example_reads[n] for n in NAMES], NAMES, [(False, 5)] * 5, False, True,
                        open(g2s, "rb").read(), True)
    assert open(outp).read() == want and out == ""
    assert "Calculated Offsets: [Forward(5), Forward(5), Forward(5), Forward(5), Forward(5)]" in err
    assert "Finished: offset_clipped; Fraction mapped: 0.999 [999 / 1000]" in err        # count.rs:34-43
    assert "Finished: diff.sequence; Fraction mapped: 0.908 [1000 / 1101]" in err
    # without -z the zero rows of zero.sequence only disappear if every sample is zero: none here
    rc, out2, _ = run(cli, "-l", LIB, "-i", os.path.join(DATA, "zero.sequence.fastq.gz"), "-a", "5", "-q")
    assert rc == 0 and out2.count("\n") == 1 + 90
    assert out2 == oracle_table(example_library_text, [example_reads["zero.sequence"]], ["zero.sequence"], [(False, 5)],
                                False, True)


def test_flags_p_r_n(cli, tmp_path, example_library_text, example_reads):
    # -p: no position recursion; -n names; reverse-complemented input with -r
    fwd = example_reads["offset_clipped"]
    lines = fwd.split(b"\n")
    rc_lines = []
    for i in range(0, len(lines) - 1, 4):
        seq = bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(lines[i + 1]))
        rc_lines += [lines[i], seq, b"+", lines[i + 3][::-1]]
    rc_text = b"\n".join(rc_lines) + b"\n"
    rp = os.path.join(str(tmp_path), "rc.fastq.gz")
    with gzip.open(rp, "wb") as f:
        f.write(rc_text)
    code, out, err = run(cli, "-l", LIB, "-i", rp, os.path.join(DATA, "offset_clipped.fastq.gz"), "-a", "5", "-r", "-p", "-q",
                         "-n", "revcomp", "forward_read_as_reverse")
    assert code == 0, err
    want = oracle_table(example_library_text, [rc_text, fwd], ["revcomp", "forward_read_as_reverse"], [(True, 5), (True, 5)],
                        False, False)
    assert out == want
    assert out.count("\n") > 50                       # the rc sample really matches


def test_error_paths(cli, tmp_path):
    seq = os.path.join(DATA, "sequence.fastq.gz")
    code, _, err = run(cli, "-l", LIB, "-i", os.path.join(str(tmp_path), "nope.fq"), "-a", "5")
    assert code == 101 and "Provided filepath does not exist" in err                 # main.rs:130-140
    code, _, err = run(cli, "-l", LIB, "-i", seq, "-n", "a", "b", "-a", "5")
    assert code == 101 and "Must provide as many sample names as there are input files" in err    # main.rs:156
    long_lib = os.path.join(str(tmp_path), "long.fa")
    open(long_lib, "wb").write(b">a\n" + b"ACGT" * 30 + b"\n")
    code, _, err = run(cli, "-l", long_lib, "-i", seq, "-a", "5", "-q")
    assert code == 1 and "Sequences in reference library are larger than the sequences in input." in err   # count.rs:98-100
    gm = os.path.join(str(tmp_path), "gm.txt")
    open(gm, "wb").write(b"gene.0\tlib.0\n")
    code, _, err = run(cli, "-l", LIB, "-i", seq, "-a", "5", "-g", gm, "-q")
    assert code == 1 and "Missing sgRNA aliases in gene map: \"lib.1\"" in err        # count.rs:90-95 (first in file order)
    code, out, _ = run(cli, "--help")
    assert code == 0 and "--library-path" in out and "--no-position-recursion" in out


@pytest.mark.parametrize("pack", ["scan", "fastq", "device", "host"])
def test_library_with_n_and_long_guides(cli, pack, tmp_path):
    """Libraries the packed records cannot carry (an 'N' inside a guide; 34-base guides) go through the byte-string path
    (sgc_bytes.h) whatever --pack says: same table as the oracle, which compares bytes like the reference does."""
    import random
    rng = random.Random(8)
    for L, alpha in ((20, b"ACGTN"), (34, b"ACGT")):
        guides = list({bytes(rng.choice(alpha) for _ in range(L)) for _ in range(300)})
        if L == 20:
            guides[0] = b"ACGTNCGTACGTACGTACGT"
        lib_text = b"".join(b">g%d\n%s\n" % (i, g) for i, g in enumerate(guides))
        reads = []
        for i in range(5000):
            g = bytearray(rng.choice(guides))
            if rng.random() < 0.3:
                g[rng.randrange(L)] = rng.choice(b"ACGTN")
            pre = bytes(rng.choice(b"ACGT") for _ in range(6 + rng.choice([0, 0, 1, -1])))
            r = pre + bytes(g) + b"GTTTTAGAGC"
            reads.append(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
        text = b"".join(reads)
        lp, fq = os.path.join(str(tmp_path), "lib%d.fa" % L), os.path.join(str(tmp_path), "s%d.fastq" % L)
        open(lp, "wb").write(lib_text)
        open(fq, "wb").write(text)
        rc, out, err = run(cli, "-l", lp, "-i", fq, "-a", "6", "-q", "--pack", pack)
        assert rc == 0, err
        assert out == oracle_table(lib_text, [text], ["s%d" % L], [(False, 6)], False, True)
        assert out.count("\n") > 100
        rc, out, err = run(cli, "-l", lp, "-i", fq, "-a", "6", "-q", "-x", "-p", "--pack", pack)      # exact, no recursion
        assert rc == 0, err
        assert out == oracle_table(lib_text, [text], ["s%d" % L], [(False, 6)], True, False)


def test_python_count_wrapper(tmp_path, example_library_text, example_reads):
    """sgcount_amd.hostlib.count(): the reference's count() signature over the in-process CLI."""
    from sgcount_amd import hostlib
    outp = os.path.join(str(tmp_path), "t.tsv")
    hostlib.count(LIB, [os.path.join(DATA, "sequence.fastq.gz"), os.path.join(DATA, "zero.sequence.fastq.gz")],
                  sample_names=["a", "b"], output_path=outp, offset=5, exact=False, include_zero=True)
    want = oracle_table(example_library_text, [example_reads["sequence"], example_reads["zero.sequence"]], ["a", "b"],
                        [(False, 5)] * 2, False, True, None, True)
    assert open(outp).read() == want
    with pytest.raises(hostlib.HostError) as e:
        hostlib.count(LIB, [os.path.join(str(tmp_path), "missing.fq")], offset=5)
    assert e.value.code == 101


def test_stats_json_and_plain_text_pipeline(cli, tmp_path, example_library_text, example_reads):
    """The two streaming paths on a plain (uncompressed) file, with tiny units of work and several threads: the default host
    scan (FastqScanner -> packed records -> sgc_sample_push_packed_async) and the GPU-parsed text (--pack fastq: tiny slices,
    several reader threads) give the oracle's table, and --stats-json reports the stages of each."""
    import json
    text = example_reads["diff.sequence"]
    fq = os.path.join(str(tmp_path), "diff.fastq")
    open(fq, "wb").write(text)
    stats = os.path.join(str(tmp_path), "stats.json")
    want = oracle_table(example_library_text, [text], ["diff"], [(False, 5)], False, True)
    rc, out, err = run(cli, "-l", LIB, "-i", fq, "-a", "5", "-q", "--scan-threads", "3", "--scan-block-kb", "4", "--stats-json", stats)
    assert rc == 0, err
    assert out == want
    st = json.load(open(stats))
    s0 = st["samples"][0]
    assert s0["scan_path"] and not s0["text_path"] and not s0["gz"] and s0["reads"] == 1101 and s0["reader_threads"] == 3
    assert s0["text_bytes"] == len(text) and st["total_s"] > 0 and s0["count_kernels_ms"] > 0 and s0["h2d_ms"] > 0
    rc, out, err = run(cli, "-l", LIB, "-i", fq, "-a", "5", "-q", "--pack", "fastq", "--io-threads", "3", "--chunk-mb", "0", "--stats-json", stats)
    assert rc == 0, err
    assert out == want
    st = json.load(open(stats))
    s0 = st["samples"][0]
    assert s0["text_path"] and not s0["scan_path"] and not s0["gz"] and s0["reads"] == 1101 and s0["reader_threads"] == 3
    assert s0["text_bytes"] == len(text) and st["total_s"] > 0 and s0["ingest_kernels_ms"] > 0 and s0["count_kernels_ms"] > 0
    # reverse strand, no recursion, several samples over two worker threads: scan path == GPU-parsed text == oracle
    fq2 = os.path.join(str(tmp_path), "seq.fastq")
    open(fq2, "wb").write(example_reads["sequence"])
    outs = []
    for pack in ("scan", "fastq"):
        rc, out, err = run(cli, "-l", LIB, "-i", fq, fq2, "-a", "5", "-r", "-p", "-q", "-t", "2", "--pack", pack, "-z")
        assert rc == 0, err
        outs.append(out)
    assert outs[0] == outs[1] == oracle_table(example_library_text, [text, example_reads["sequence"]], ["diff", "seq"],
                                              [(True, 5), (True, 5)], False, False, include_zero=True)


def test_bgzf_input_is_inflated_in_parallel(cli, tmp_path, example_library_text, example_reads):
    """A BGZF-compressed FASTQ (gzip members with a 'BC' size field) goes through the multi-threaded inflate of the text
    path: same table as the oracle on the plain text; --stats-json says so."""
    import json
    from sgcount_amd.bgzf import bgzf_bytes
    text = example_reads["diff.sequence"] * 40
    p = os.path.join(str(tmp_path), "diff.fastq.gz")
    open(p, "wb").write(bgzf_bytes(text, block=20000))
    stats = os.path.join(str(tmp_path), "stats.json")
    # default: the host scan inflates the members and packs (sgh_scan.cpp run_gz); --pack fastq: the text path inflates them into
    # pinned slices and the GPU parses
    for pack, key in (("scan", "scan_path"), ("fastq", "text_path")):
        rc, out, err = run(cli, "-l", LIB, "-i", p, "-a", "5", "-q", "--io-threads", "4", "--scan-threads", "4", "--chunk-mb", "0", "--pack", pack,
                           "--stats-json", stats)
        assert rc == 0, err
        assert out == oracle_table(example_library_text, [text], ["diff"], [(False, 5)], False, True)
        s0 = json.load(open(stats))["samples"][0]
        assert s0[key] and s0["gz"] and s0["bgzf"] and s0["reads"] == 1101 * 40, (pack, s0)
        assert s0["reader_threads"] == 4 if pack == "fastq" else 1 <= s0["reader_threads"] <= 4      # (the scanner: as many as the file has chunks)
    # a corrupt member is an error, not a silent miscount
    blob = bytearray(open(p, "rb").read())
    blob[len(blob) // 2] ^= 0x55
    open(p, "wb").write(bytes(blob))
    rc, out, err = run(cli, "-l", LIB, "-i", p, "-a", "5", "-q")
    assert rc != 0 and ("BGZF" in err or "read error" in err)      # whichever reader meets the bad member first


def test_plain_gzip_input_is_inflated_in_parallel(cli, tmp_path, example_library_text, example_reads):
    """An ordinary .gz (one deflate stream, what `gzip` and the sequencers write and every example of the reference is) goes
    through the multi-threaded speculative inflater of the text path (sgh_inflate.cpp): same table as the oracle on the plain
    text, several gzip members included; --stats-json says so; a corrupt stream is an error."""
    import json
    text = example_reads["diff.sequence"] * 60
    want = oracle_table(example_library_text, [text], ["diff"], [(False, 5)], False, True)
    stats = os.path.join(str(tmp_path), "stats.json")
    third = len(text) // 3
    third -= third % 4          # (any byte position is fine: members are concatenated text)
    for name, blob in (("one.fastq.gz", gzip.compress(text, 6)),
                       ("three.fastq.gz", gzip.compress(text[:third], 1) + gzip.compress(text[third:2 * third], 9) + gzip.compress(text[2 * third:], 6))):
        p = os.path.join(str(tmp_path), name)
        open(p, "wb").write(blob)
        # default: the host scan inflates and packs (sgh_scan.cpp run_gz); --pack fastq: the text path inflates into pinned slices and
        # the GPU parses (TextFeeder::run_pgz) — the same speculative decoder under both
        for pack, key in (("scan", "scan_path"), ("fastq", "text_path")):
            rc, out, err = run(cli, "-l", LIB, "-i", p, "-a", "5", "-q", "-n", "diff", "--io-threads", "4", "--scan-threads", "4", "--chunk-mb", "1",
                               "--pack", pack, "--stats-json", stats)
            assert rc == 0, err
            assert out == want
            s0 = json.load(open(stats))["samples"][0]
            assert s0[key] and s0["gz"] and s0["parallel_gzip"] and not s0["bgzf"] and s0["reads"] == 1101 * 60, (pack, s0)
    blob = bytearray(gzip.compress(text, 6))
    blob[len(blob) // 2] ^= 0x41
    p = os.path.join(str(tmp_path), "bad.fastq.gz")
    open(p, "wb").write(bytes(blob))
    for pack in ("scan", "fastq"):
        rc, out, err = run(cli, "-l", LIB, "-i", p, "-a", "5", "-q", "--pack", pack)
        assert rc != 0 and ("gzip" in err or "read error" in err or "panicked" in err), (pack, err)


def test_malformed_fastq_panics_like_the_reference(cli, tmp_path, example_reads):
    """fxread panics on a malformed record (unpinned): exit code 101, on the text path (GPU-verified marker
    bytes) and on the record-reader path alike; a truncated last record too."""
    text = example_reads["sequence"]
    lines = text.split(b"\n")
    bad = list(lines)
    bad[4 * 500 + 2] = b"-"                         # separator line without '+'
    for name, body in (("bad_plus.fastq", b"\n".join(bad)), ("trunc.fastq", b"\n".join(lines[: 4 * 700 + 2]) + b"\n")):
        p = os.path.join(str(tmp_path), name)
        open(p, "wb").write(body)
        for pack in ("scan", "fastq", "device"):
            rc, out, err = run(cli, "-l", LIB, "-i", p, "-a", "5", "-q", "--pack", pack)
            assert rc == 101, (name, pack, rc, err)
            assert "panicked" in err


@pytest.mark.parametrize("kind", ["plain", "gz", "bgzf"])
def test_fastq_endings_golden(cli, tmp_path, kind):
    """How a FASTQ stream may end (reader decision #3; tests/golden/fastq_endings.json, generated from the oracle): a
    stream that ends behind a separator line ends with a record whose quality line is empty; blank lines at the very end are not
    records; every other incomplete tail is refused with the reference's panic exit code — through every reader of the command
    line (--pack scan / fastq / device / host), on plain text, one gzip stream and BGZF."""
    import json
    from sgcount_amd import bgzf
    g = json.load(open(os.path.join(os.path.dirname(DATA), "golden", "fastq_endings.json")))
    lib = os.path.join(str(tmp_path), "lib.fa")
    open(lib, "w").write(g["library"])
    ids = [ln[1:] for ln in g["library"].split("\n") if ln.startswith(">")]
    good, bad = [], []
    for k, c in enumerate(g["cases"]):
        text = (g["body"] + c["tail"]).encode()
        p = os.path.join(str(tmp_path), "c%02d.fastq%s" % (k, "" if kind == "plain" else ".gz"))
        open(p, "wb").write(text if kind == "plain" else gzip.compress(text) if kind == "gz" else bgzf.bgzf_bytes(text, block=700))
        (bad if c.get("error") else good).append((p, c))
    names = ["s%d" % i for i in range(len(good))]
    want = "Guide\t" + "\t".join(names) + "\n" + "".join(
        "%s\t%s\n" % (ident, "\t".join(str(c["counts"][j]) for _, c in good)) for j, ident in enumerate(ids) if any(c["counts"][j] for _, c in good))
    for pack in ("scan", "fastq", "device", "host"):
        rc, out, err = run(cli, "-l", lib, "-i", *[p for p, _ in good], "-n", *names, "-a", str(g["offset"]), "--pack", pack)
        assert rc == 0, (kind, pack, err)
        assert out == want, (kind, pack)
        for (p, c), name in zip(good, names):
            assert "Finished: %s; Fraction mapped: %.3f [%d / %d]" % (name, c["matched"] / c["total"], c["matched"], c["total"]) in err, (kind, pack, name, err)
    for pack in ("scan", "fastq", "host"):
        for p, c in bad:
            rc, out, err = run(cli, "-l", lib, "-i", p, "-a", str(g["offset"]), "-q", "--pack", pack)
            assert rc == 101 and "panicked" in err, (kind, pack, c["tail"], rc, err)
