"""GPU: the streaming FASTQ ingest (sgc_sample_push_fastq_part, sgc_fastq.hip) against the oracle fed the same text.

Covers what the reference's fxread iterator covers inside Counter::count (src/counter.rs:211-236, call site
src/count.rs:24): parts cut at arbitrary LINE boundaries (any phase of the 4-line cycle), forward and reverse strand,
one- and two-word records, thousands of tiny records per tile (list overflow), marker-byte validation ('@' / '+'),
CRLF terminators, offsets larger than the staged halo, a wrong announced newline count."""
import ctypes as C
import random

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import sgcount_amd as S
    from sgcount_amd import synth, workload
    S._ffi.load()
    return torch, S, synth, workload


def _count_parts(torch, S, dl, parts, reverse, offset, recursion, where, announce=True, finish_rc=0):
    """Pushes `parts` (bytes objects forming one FASTQ stream, each a run of whole lines) and returns the table."""
    ffi = S._ffi
    smp = C.c_void_p()
    ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), int(reverse), offset, int(recursion)))
    first_line, total, keep = 0, 0, []
    try:
        for p in parts:
            nl = p.count(b"\n")
            nrec = C.c_uint64(0)
            ann = C.c_uint64(nl if announce else 2 ** 64 - 1)
            if where == ffi.MEM_DEVICE:
                d = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
                keep.append(d)
                ffi.check(dl.lib.sgc_sample_push_fastq_part(smp, d.data_ptr(), len(p), where, first_line, ann, C.byref(nrec)))
            else:
                buf = C.create_string_buffer(p, len(p))
                keep.append(buf)
                ffi.check(dl.lib.sgc_sample_push_fastq_part(smp, buf, len(p), where, first_line, ann, C.byref(nrec)))
            total += nrec.value
            first_line += nl + (0 if p.endswith(b"\n") else 1)
        ffi.check(dl.lib.sgc_sample_wait_uploads(smp, 0))
        out = np.zeros(len(dl.library), dtype=np.uint64)
        t, m = C.c_uint64(), C.c_uint64()
        rc = dl.lib.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m))
        assert rc == finish_rc, (rc, dl.lib.sgc_last_error())
        if rc == 0:
            assert total == t.value
        return out.tolist(), t.value, m.value
    finally:
        dl.lib.sgc_sample_free(smp)


def _cut_at_lines(text, rng, n_cuts):
    nl = [i for i, c in enumerate(text) if c == 10]
    cuts = sorted(set(rng.sample(nl, min(n_cuts, len(nl)))))
    parts, prev = [], 0
    for c in cuts:
        parts.append(text[prev:c + 1]); prev = c + 1
    if prev < len(text):
        parts.append(text[prev:])
    return [p for p in parts if p]


@pytest.mark.parametrize("L,reverse,recursion", [(20, False, True), (20, True, True), (20, True, False), (27, False, True), (27, True, True)])
def test_parts_at_any_line_vs_oracle(env, L, reverse, recursion):
    torch, S, synth, workload = env
    ffi = S._ffi
    n, ng = 50_000, 3000
    lib_seqs, library = workload.synth_library(ng, L)
    text = synth.fastq_host(lib_seqs, 4242, n, mode=1)
    lines = text.split(b"\n")
    rng = np.random.default_rng(9)
    for i in rng.integers(0, n, 3000):
        s = bytearray(lines[4 * i + 1])
        if s:
            s[rng.integers(0, len(s))] = rng.choice(np.frombuffer(b"NNNnRJ", dtype=np.uint8))
            lines[4 * i + 1] = bytes(s)
    lines[4 * 5 + 1] = b""; lines[4 * 5 + 3] = b""                    # an empty read
    lines[4 * 9 + 1] = b"ACGT"; lines[4 * 9 + 3] = b"IIII"            # a read shorter than the offset
    text = b"\n".join(lines)
    lib_text = synth.library_fasta(lib_seqs)
    want = O.count_text(lib_text, text, reverse, 30, False, recursion)
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    r = random.Random(5 + L + reverse)
    assert _count_parts(torch, S, dl, [text], reverse, 30, recursion, ffi.MEM_HOST) == want
    assert _count_parts(torch, S, dl, [text[:-1]], reverse, 30, recursion, ffi.MEM_DEVICE) == want       # no final newline
    for n_cuts in (1, 3, 40):
        parts = _cut_at_lines(text, r, n_cuts)
        assert {p.count(b"\n") % 4 for p in parts} != {0} or n_cuts == 1            # parts really start at every phase
        assert _count_parts(torch, S, dl, parts, reverse, 30, recursion, ffi.MEM_HOST) == want
        assert _count_parts(torch, S, dl, parts, reverse, 30, recursion, ffi.MEM_DEVICE) == want
    # the newline count left to the device (one synchronisation per part)
    assert _count_parts(torch, S, dl, _cut_at_lines(text, r, 5), reverse, 30, recursion, ffi.MEM_HOST, announce=False) == want


@pytest.mark.parametrize("reverse", [False, True])
def test_thousands_of_tiny_records(env, reverse):
    """8-12 byte records: a 64 KiB tile holds up to 8192 of them, far more than one listing round (the round-1
    kernel overflowed its 1024-entry LDS list here)."""
    torch, S, synth, workload = env
    ffi = S._ffi
    rng = random.Random(12)
    guides = [b"ACG", b"TTA", b"GGC", b"CAT", b"AAA"]
    lib_text = b"".join(b">g%d\n%s\n" % (i, g) for i, g in enumerate(guides))
    library = S.Library.from_reader(S.parse_fastx(lib_text))
    recs = []
    for i in range(60_000):
        k = rng.choice([1, 2, 3, 3, 3, 4, 5])
        seq = bytes(rng.choice(b"ACGTN") for _ in range(k)) if rng.random() < 0.5 else rng.choice(guides)[:k]
        recs.append(b"@\n%s\n+\n%s\n" % (seq, b"I" * len(seq)))
    text = b"".join(recs)
    assert len(text) / len(recs) < 12.5
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    for offset in (0, 1):
        want = O.count_text(lib_text, text, reverse, offset, False, True)
        assert _count_parts(torch, S, dl, [text], reverse, offset, True, ffi.MEM_DEVICE) == want
        assert _count_parts(torch, S, dl, _cut_at_lines(text, rng, 7), reverse, offset, True, ffi.MEM_HOST) == want
    assert want[1] > 50_000 and want[2] > 1000


def test_marker_bytes_are_verified(env):
    """A header line that does not start with '@' or a separator that does not start with '+' is a malformed record:
    fxread panics (unpinned); here sgc_sample_finish returns SGC_E_FORMAT naming the line."""
    torch, S, synth, workload = env
    ffi = S._ffi
    lib_seqs, library = workload.synth_library(500, 20)
    text = synth.fastq_host(lib_seqs, 0, 3000)
    lines = text.split(b"\n")
    dl = library.device(False)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    good = _count_parts(torch, S, dl, [text], False, 30, True, ffi.MEM_DEVICE)
    assert good[1] == 3000
    # quality lines full of '@' and '+' are fine
    ok = list(lines)
    for r in (0, 17, 2999):
        ok[4 * r + 3] = b"@+" * (len(ok[4 * r + 3]) // 2) + b"@" * (len(ok[4 * r + 3]) % 2)
    assert _count_parts(torch, S, dl, [b"\n".join(ok)], False, 30, True, ffi.MEM_DEVICE) == good
    for rec, line_in_rec, byte in ((0, 0, b"r"), (1234, 0, b">"), (2999, 2, b"-"), (700, 2, b"@"), (1, 0, b"+")):
        bad = list(lines)
        bad[4 * rec + line_in_rec] = byte + bad[4 * rec + line_in_rec][1:]
        for parts in ([b"\n".join(bad)], _cut_at_lines(b"\n".join(bad), random.Random(rec), 9)):
            for where in (ffi.MEM_DEVICE, ffi.MEM_HOST):
                _count_parts(torch, S, dl, parts, False, 30, True, where, finish_rc=ffi.E_FORMAT)
                msg = dl.lib.sgc_last_error().decode()
                assert "line %d " % (4 * rec + line_in_rec + 1) in msg, msg
    # a dropped line shifts every later record: reported at the first line that breaks the cycle
    dropped = lines[:4 * 100 + 1] + lines[4 * 100 + 2:]
    _count_parts(torch, S, dl, [b"\n".join(dropped)], False, 30, True, ffi.MEM_DEVICE, finish_rc=ffi.E_FORMAT)
    # a wrong announced newline count is caught too
    smp = C.c_void_p()
    ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, 30, 1))
    d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    ffi.check(dl.lib.sgc_sample_push_fastq_part(smp, d.data_ptr(), len(text), ffi.MEM_DEVICE, 0, text.count(b"\n") - 4, None))
    assert dl.lib.sgc_sample_finish(smp, None, None, None) == ffi.E_FORMAT
    dl.lib.sgc_sample_free(smp)


def test_crlf_terminators(env):
    """CRLF text counts like LF text (the '\\r' belongs to the terminator) — a decision, not a reference fact."""
    torch, S, synth, workload = env
    ffi = S._ffi
    lib_seqs, library = workload.synth_library(800, 20)
    text = synth.fastq_host(lib_seqs, 100, 5000, mode=1)
    # reads that END exactly where the Centered / Plus window ends: a kept '\r' would move the bounds
    lines = text.split(b"\n")
    for r in range(0, 5000, 7):
        cut = 50 + (r % 3)
        lines[4 * r + 1] = lines[4 * r + 1][:cut]; lines[4 * r + 3] = lines[4 * r + 3][:cut]
    text = b"\n".join(lines)
    crlf = text.replace(b"\n", b"\r\n")
    lib_text = synth.library_fasta(lib_seqs)
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    for reverse in (False, True):
        want = O.count_text(lib_text, text, reverse, 30, False, True)
        assert O.count_text(lib_text, crlf, reverse, 30, False, True) == want
        assert _count_parts(torch, S, dl, [crlf], reverse, 30, True, ffi.MEM_DEVICE) == want
        assert _count_parts(torch, S, dl, _cut_at_lines(crlf, random.Random(3), 11), reverse, 30, True, ffi.MEM_HOST) == want


@pytest.mark.parametrize("reverse", [False, True])
def test_offset_beyond_the_staged_halo(env, reverse):
    """offset + L + 2 > 1 KiB: the owning lanes read the line from global memory instead of the LDS halo."""
    torch, S, synth, workload = env
    ffi = S._ffi
    rng = random.Random(77)
    guides = sorted({bytes(rng.choice(b"ACGT") for _ in range(12)) for _ in range(200)})
    lib_text = b"".join(b">g%d\n%s\n" % (i, g) for i, g in enumerate(guides))
    library = S.Library.from_reader(S.parse_fastx(lib_text))
    o = 1500
    recs = []
    for i in range(400):
        g = rng.choice(guides)
        pre = bytes(rng.choice(b"ACGT") for _ in range(o + rng.choice([0, 0, 1, -1])))
        tail = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 1, 2, 700])))
        seq = pre + g + tail
        if reverse:
            seq = bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(seq))
        if i % 50 == 0:
            seq = seq[:rng.randrange(len(seq))]
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, seq, b"I" * len(seq)))
    text = b"".join(recs)
    want = O.count_text(lib_text, text, reverse, o, False, True)
    assert want[2] > 300
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    assert _count_parts(torch, S, dl, [text], reverse, o, True, ffi.MEM_DEVICE) == want
    assert _count_parts(torch, S, dl, _cut_at_lines(text, rng, 6), reverse, o, True, ffi.MEM_HOST) == want


def test_a_stream_may_end_behind_a_separator_line(env):
    """Reader decision #3 (tests/golden/fastq_endings.json): the last record's quality line may be empty and unterminated.
    The whole-text entry point takes such a text (3 lines of a record at its end), the part-wise one counts its sequence line."""
    import json
    import os
    torch, S, synth, workload = env
    ffi = S._ffi
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fastq_endings.json")))
    library = S.Library.from_reader(S.parse_fastx(g["library"].encode()))
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    for c in g["cases"]:
        text = (g["body"] + c["tail"]).encode()
        while text.endswith(b"\n\n"):                 # blank lines at the very end are the host's to cut (TextFeeder, FastqScanner)
            text = text[:-1]
        lines = text.count(b"\n") + (0 if text.endswith(b"\n") else 1)
        if c.get("error"):
            # 1 or 2 lines of a record at the end: the push refuses the text; a whole number of lines with a wrong marker byte: the finish does
            smp = C.c_void_p()
            ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, g["offset"], 1))
            buf = C.create_string_buffer(text, len(text))
            rc_push = dl.lib.sgc_sample_push_fastq(smp, buf, len(text), ffi.MEM_HOST, None)
            out = np.zeros(len(library), dtype=np.uint64)
            t, m = C.c_uint64(), C.c_uint64()
            rc_fin = dl.lib.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m))
            assert (rc_push != 0) == (lines % 4 in (1, 2)) and (rc_push != 0 or rc_fin == ffi.E_FORMAT), (c["tail"], rc_push, rc_fin)
            dl.lib.sgc_sample_free(smp)
            continue
        want = (c["counts"], c["total"], c["matched"])
        assert _count_parts(torch, S, dl, [text], False, g["offset"], True, ffi.MEM_DEVICE) == want, c["tail"]
        assert _count_parts(torch, S, dl, _cut_at_lines(text, random.Random(5), 7), False, g["offset"], True, ffi.MEM_HOST) == want, c["tail"]
        smp = C.c_void_p()
        ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, g["offset"], 1))
        try:
            buf = C.create_string_buffer(text, len(text))
            nrec = C.c_uint64(0)
            ffi.check(dl.lib.sgc_sample_push_fastq(smp, buf, len(text), ffi.MEM_HOST, C.byref(nrec)))
            out = np.zeros(len(library), dtype=np.uint64)
            t, m = C.c_uint64(), C.c_uint64()
            ffi.check(dl.lib.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m)))
            assert (out.tolist(), t.value, m.value) == want and nrec.value == c["total"], c["tail"]
        finally:
            dl.lib.sgc_sample_free(smp)
