"""GPU: synthetic generator host == device, device packer == host packer, and the synthetic
workload counted on the GPU == the CPU oracle on the same reads (bit-exact)."""
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


@pytest.mark.parametrize("mode", [0, 1])
def test_generator_device_matches_host(env, mode):
    torch, S, synth, _ = env
    lib = synth.library(3000, 20)
    lib_dev = torch.from_numpy(lib).cuda()
    raw, offs = synth.reads_device(lib_dev, 12345, 50000, mode=mode)
    h_raw, h_offs = synth.reads_host(lib, 12345, 50000, mode=mode)
    assert np.array_equal(offs.cpu().numpy().astype(np.uint64), h_offs)
    assert np.array_equal(raw.cpu().numpy(), h_raw)
    fq, _ = synth.fastq_device(lib_dev, 999_999_990, 2000, mode=mode)      # crosses a digit-count boundary
    assert fq.cpu().numpy().tobytes() == synth.fastq_host(lib, 999_999_990, 2000, mode=mode)


@pytest.mark.parametrize("L,reverse,recursion", [(20, False, True), (20, True, True), (20, False, False), (27, False, True)])
def test_device_packer_matches_host_packer(env, L, reverse, recursion):
    torch, S, synth, workload = env
    import ctypes as C
    lib_seqs, library = workload.synth_library(2000, L)
    h_raw, h_offs = synth.reads_host(lib_seqs, 0, 40000, mode=1)
    reads = [h_raw[int(h_offs[i]):int(h_offs[i + 1])].tobytes() for i in range(40000)]
    # sprinkle non-ACGT bytes
    rng = np.random.default_rng(3)
    raw = h_raw.copy()
    idx = rng.integers(0, len(raw), 4000)
    raw[idx] = rng.choice(np.frombuffer(b"NNNnRJ", dtype=np.uint8), 4000)
    reads = [raw[int(h_offs[i]):int(h_offs[i + 1])].tobytes() for i in range(40000)]
    off = S.Offset.Reverse(30) if reverse else S.Offset.Forward(30)
    want = S.pack_reads_host(reads, L, off, recursion)
    dl = library.device(False)
    d_raw = torch.from_numpy(raw).cuda()
    d_offs = torch.from_numpy(h_offs.astype(np.int64)).cuda()
    out = torch.empty(len(want), dtype=torch.int64, device="cuda")
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    S._ffi.check(dl.lib.sgc_pack_reads_device(dl.ctx, d_raw.data_ptr(), d_offs.data_ptr(), 40000, int(reverse), 30,
                                              int(recursion), out.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("exact", [False, True])
def test_synthetic_workload_vs_oracle_2m(env, exact):
    """2M reads of the bench workload (10k guides so that the hot guides are really hot) vs the oracle."""
    torch, S, synth, workload = env
    n, ng = 2_000_000, 10_000
    wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=not exact, gen_chunk=700_000)
    wl.step()
    counts, total, matched = wl.result()
    lib_text = synth.library_fasta(wl.lib_seqs)
    lib = O.Library(lib_text)
    ctr = O.Counter(lib, None if exact else O.Permuter(lib), False, 30, 20, True)
    for first in range(0, n, 500_000):
        ctr.feed_text(synth.fastq_host(wl.lib_seqs, first, 500_000))
    assert total == n == ctr.total_reads()
    assert matched == ctr.matched_reads()
    assert counts.tolist() == ctr.table()
    # linearity: two half passes accumulate to the same table
    S._ffi.check(wl.abi.sgc_sample_reset(wl.sample))
    for first, m in ((0, 1_234_567), (1_234_567, n - 1_234_567)):
        S._ffi.check(wl.abi.sgc_sample_push_packed(wl.sample, wl.records.data_ptr() + 8 * first, m, S._ffi.MEM_DEVICE))
    S._ffi.check(wl.abi.sgc_sample_export_device(wl.sample, wl.export.data_ptr()))
    c2, t2, m2 = wl.result()
    assert (c2.tolist(), t2, m2) == (counts.tolist(), total, matched)
    wl.close()


@pytest.mark.parametrize("exact", [False, True])
def test_every_fallback_mode_gives_the_same_table(env, exact):
    """The shipped pass is a stack of results-preserving choices (balanced shares of the slice blocks, five-byte > six-byte slice blocks > direct miss runs > dense runs >
    sub-partition tag > two-choice image > core-hashed slices); each one has a fallback that libraries of other shapes take
    (L >= 22, slices that do not follow the core hash, ...).  Every rung of that ladder must count the same table — the
    oracle's — on the same 1M reads."""
    torch, S, synth, workload = env
    n, ng = 1_000_000, 20_000
    lib_text = None
    want = None
    ladder = [({}, {}), ({}, {"five_byte": 0}), ({}, {"five_byte": 0, "six_byte": 0}), ({}, {"balanced": 0}), ({}, {"direct": 0}), ({}, {"dense": 0}), ({}, {"tag_sub": 0}), ({}, {"cuckoo": 0}),
              ({}, {"direct": 0, "cuckoo": 0, "tag_sub": 0}), ({"align_slices": 0}, {}), ({"align_slices": 0}, {"dense": 0, "cuckoo": 0}),
              ({"rest_filter": 0}, {}), ({}, {"variant": 3}), ({}, {"variant": 1})]
    for lib_opts, opts in ladder:
        wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=not exact, gen_chunk=500_000, lib_options=lib_opts)
        for k, v in opts.items():
            wl.dl.set_option(k, v)
        wl.step()
        counts, total, matched = wl.result()
        if want is None:
            lib_text = synth.library_fasta(wl.lib_seqs)
            lib = O.Library(lib_text)
            ctr = O.Counter(lib, None if exact else O.Permuter(lib), False, 30, 20, True)
            for first in range(0, n, 500_000):
                ctr.feed_text(synth.fastq_host(wl.lib_seqs, first, 500_000))
            want = (ctr.table(), ctr.total_reads(), ctr.matched_reads())
        assert (counts.tolist(), total, matched) == want, (lib_opts, opts)
        wl.close()


@pytest.mark.parametrize("pct", [10, 60])
def test_sample_dominated_by_one_guide_vs_oracle(env, pct):
    """pct percent of the reads draw ONE guide (synth.mode_dominant): its slice holds most of the slice blocks, so with balanced
    shares most workgroups of k_count_slices work on that one slice (and some on two or three slices: segments), with static
    shares eight of them do.  Both must count the oracle's table."""
    torch, S, synth, workload = env
    n, ng = 1_000_000, 20_000
    mode = synth.MODE_FIXED | synth.mode_dominant(pct)
    want = None
    for opts in ({}, {"balanced": 0}):
        wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=True, gen_chunk=500_000, mode=mode)
        for k, v in opts.items():
            wl.dl.set_option(k, v)
        wl.step()
        counts, total, matched = wl.result()
        if want is None:
            lib = O.Library(synth.library_fasta(wl.lib_seqs))
            ctr = O.Counter(lib, O.Permuter(lib), False, 30, 20, True)
            for first in range(0, n, 500_000):
                ctr.feed_text(synth.fastq_host(wl.lib_seqs, first, 500_000, mode=mode))
            want = (ctr.table(), ctr.total_reads(), ctr.matched_reads())
            assert max(want[0]) >= pct * n * 0.85 // 100            # the dominant guide really dominates
        assert (counts.tolist(), total, matched) == want, opts
        wl.close()


@pytest.mark.parametrize("variant,ng", [(4, 150_000), (3, 150_000), (4, 200_000)])
def test_large_library_128_slices(env, variant, ng):
    """150k-200k guides need 128 library slices (64 hold ~105k) and, from ~170k, 512 core partitions: still the
    partitioned path with the in-LDS resolver, still the oracle's table."""
    torch, S, synth, workload = env
    n = 1_500_000
    wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=True, gen_chunk=700_000)
    info = wl.dl.info()
    assert info.lib_slots == 1 << 19 and info.core_partitions == (256 if ng == 150_000 else 512)
    assert info.path == 4 and info.slices == 128 and info.slice_record_bytes == 5
    wl.dl.set_option("variant", variant)
    wl.step()
    counts, total, matched = wl.result()
    lib_text = synth.library_fasta(wl.lib_seqs)
    lib = O.Library(lib_text)
    ctr = O.Counter(lib, O.Permuter(lib), False, 30, 20, True)
    for first in range(0, n, 500_000):
        ctr.feed_text(synth.fastq_host(wl.lib_seqs, first, 500_000))
    assert (counts.tolist(), total, matched) == (ctr.table(), ctr.total_reads(), ctr.matched_reads())
    # the partitioned kernels ran (the generic fallback has no partition stage)
    wl.dl.timing(True); wl.dl.timing(reset=True)
    wl.step()
    torch.cuda.synchronize()
    assert wl.dl.timing(reset=True).part_ms > 0
    wl.dl.timing(False)
    wl.close()


@pytest.mark.parametrize("ng", [250_000, 400_000])
def test_large_library_big_slices(env, ng):
    """Libraries beyond 128 slices of 2^12 slots (> ~210k guides: tiling / paired-guide libraries) get slices of 2^13 slots — one
    workgroup of k_count_slices per CU — and stay on the partitioned pass (round 3: the generic kernels).  250k guides: 128 big slices
    + the in-LDS core resolver (512 core partitions); 400k guides: the core index no longer fits 512 partitions of 2048 entries, so the
    probing resolver (path 3) takes the leftovers.  The oracle's table either way, for the default and for the fallback rungs."""
    torch, S, synth, workload = env
    n = 1_500_000
    wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=True, gen_chunk=700_000)
    info = wl.dl.info()
    assert info.lib_slots == 1 << 20 and info.slices == 128
    assert info.path == (4 if ng == 250_000 else 3), info.path
    lib_text = synth.library_fasta(wl.lib_seqs)
    lib = O.Library(lib_text)
    ctr = O.Counter(lib, O.Permuter(lib), False, 30, 20, True)
    for first in range(0, n, 500_000):
        ctr.feed_text(synth.fastq_host(wl.lib_seqs, first, 500_000))
    want = (ctr.table(), ctr.total_reads(), ctr.matched_reads())
    for opts in ({}, {"cuckoo": 0}, {"five_byte": 0}, {"variant": 3}, {"variant": 1}):
        for k, v in opts.items():
            wl.dl.set_option(k, v)
        wl.step()
        counts, total, matched = wl.result()
        assert (counts.tolist(), total, matched) == want, opts
        for k in opts:
            wl.dl.set_option(k, 4 if k == "variant" else 1)
    wl.dl.timing(True); wl.dl.timing(reset=True)
    wl.step()
    torch.cuda.synchronize()
    assert wl.dl.timing(reset=True).part_ms > 0            # the partitioned kernels ran
    wl.dl.timing(False)
    wl.close()


@pytest.mark.parametrize("L,reverse,recursion", [(20, False, True), (20, True, False), (27, False, True)])
def test_fastq_ingest_on_device(env, L, reverse, recursion):
    """sgc_sample_push_fastq: record boundaries + packing on the GPU from raw FASTQ text == the oracle fed the
    same text; chunked pushes (record-aligned), device- and host-resident text, missing final newline."""
    import ctypes as C
    torch, S, synth, workload = env
    ffi = S._ffi
    n, ng = 60_000, 3000
    lib_seqs, library = workload.synth_library(ng, L)
    text = synth.fastq_host(lib_seqs, 777, n, mode=1)
    # sprinkle non-ACGT bytes into sequence lines only (line 1 of every record)
    lines = text.split(b"\n")
    rng = np.random.default_rng(9)
    for i in rng.integers(0, n, 3000):
        s = bytearray(lines[4 * i + 1])
        if s:
            s[rng.integers(0, len(s))] = rng.choice(np.frombuffer(b"NNNnRJ", dtype=np.uint8))
            lines[4 * i + 1] = bytes(s)
    lines[4 * 5 + 1] = b""                       # an empty read
    lines[4 * 5 + 3] = b""
    text = b"\n".join(lines)
    lib_text = synth.library_fasta(lib_seqs)
    want, tot, mat = O.count_text(lib_text, text, reverse, 30, False, recursion)
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)

    def run(chunks, where):
        smp = C.c_void_p()
        ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), int(reverse), 30, int(recursion)))
        total = 0
        keep = []
        for ch in chunks:
            nrec = C.c_uint64(0)
            if where == ffi.MEM_DEVICE:
                d = torch.frombuffer(bytearray(ch), dtype=torch.uint8).cuda()
                keep.append(d)
                ffi.check(dl.lib.sgc_sample_push_fastq(smp, d.data_ptr(), len(ch), where, C.byref(nrec)))
            else:
                ffi.check(dl.lib.sgc_sample_push_fastq(smp, ch, len(ch), where, C.byref(nrec)))
            total += nrec.value
        out = np.zeros(ng, dtype=np.uint64)
        t, m = C.c_uint64(), C.c_uint64()
        ffi.check(dl.lib.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m)))
        dl.lib.sgc_sample_free(smp)
        assert total == t.value
        return out.tolist(), t.value, m.value

    assert run([text], ffi.MEM_HOST) == (want, tot, mat)
    assert run([text[:-1]], ffi.MEM_DEVICE) == (want, tot, mat)          # no trailing newline
    # record-aligned chunks of uneven size (a chunk boundary may fall anywhere relative to the 4 KiB tiles)
    cuts = [0]
    pos = 0
    for k in (1, 7, 100, 4097, 20000):
        for _ in range(4 * k):
            pos = text.index(b"\n", pos) + 1
        cuts.append(pos)
    cuts.append(len(text))
    chunks = [text[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    assert run(chunks, ffi.MEM_DEVICE) == (want, tot, mat)
    # a chunk that is not a whole number of records is rejected
    smp = C.c_void_p()
    ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, 30, 1))
    bad = text[: text.index(b"\n", 5000) + 1]
    while bad.count(b"\n") % 4 in (0, 3):       # (3 lines of a record at the end are a record with an empty quality line: DESIGN.md §2, decision #3)
        bad = bad[: bad.rindex(b"\n", 0, len(bad) - 1) + 1]
    assert dl.lib.sgc_sample_push_fastq(smp, bad, len(bad), ffi.MEM_HOST, None) == ffi.E_ARG
    dl.lib.sgc_sample_free(smp)


def test_auto_offset_stagger_workload(env, tmp_path):
    """BASELINE.json configs[3]: variable-stagger reads (prefix 28..32), no -a: the offsetter must find
    Forward(30) on the first 5000 reads, and the counts with that offset must equal the oracle's."""
    import os
    torch, S, synth, workload = env
    from sgcount_amd import hostlib
    n, ng = 300_000, 5000
    wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=True, mode=synth.MODE_STAGGER, gen_chunk=100_000)
    lib_text = synth.library_fasta(wl.lib_seqs)
    fq = synth.fastq_host(wl.lib_seqs, 0, n, mode=synth.MODE_STAGGER)
    lp, rp = os.path.join(str(tmp_path), "lib.fa"), os.path.join(str(tmp_path), "reads.fq")
    open(lp, "wb").write(lib_text)
    open(rp, "wb").write(fq)
    assert hostlib.entropy_offset_group(lp, [rp]) == [(False, 30)] == [O.entropy_offset(lib_text, fq)]
    wl.step()
    counts, total, matched = wl.result()
    want, tot, mat = O.count_text(lib_text, fq, False, 30, False, True)
    assert (counts.tolist(), total, matched) == (want, tot, mat)
    # prefix 29/31 are rescued by the position recursion, 28/32 are not: roughly 10 % of the guide reads are lost
    assert 0.80 < matched / total < 0.90
    wl.close()


def test_full_size_properties(env):
    """BASELINE.json's full size (100k guides, 100M reads) through size-independent properties: totals, the
    count-sum invariant, linearity of the fold over a split, agreement of every kernel variant, and the
    oracle on a prefix."""
    torch, S, synth, workload = env
    ffi = S._ffi
    n, ng = 100_000_000, 100_000
    wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=True)
    wl.step()
    counts, total, matched = wl.result()
    assert total == n and int(counts.sum()) == matched
    assert 0.945 < matched / total < 0.955                     # 85+5+1+2+2 % of the classes can match
    # hot guides (every 100th) carry 50x weight
    hot = counts[::100].astype(np.float64).mean()
    cold = np.delete(counts, np.arange(0, ng, 100)).astype(np.float64).mean()
    assert 45 < hot / cold < 55
    # linearity: two pushes accumulate to the same table
    ffi.check(wl.abi.sgc_sample_reset(wl.sample))
    cut = 37_000_001
    for first, m in ((0, cut), (cut, n - cut)):
        ffi.check(wl.abi.sgc_sample_push_packed(wl.sample, wl.records.data_ptr() + 8 * first, m, ffi.MEM_DEVICE))
    ffi.check(wl.abi.sgc_sample_export_device(wl.sample, wl.export.data_ptr()))
    c2, t2, m2 = wl.result()
    assert t2 == total and m2 == matched and np.array_equal(c2, counts)
    # every variant of the count path gives the same table
    for v in (1, 3):
        wl.dl.set_option("variant", v)
        wl.step()
        cv, tv, mv = wl.result()
        assert tv == total and mv == matched and np.array_equal(cv, counts), v
    wl.dl.set_option("variant", 4)
    # small internal chunks (several pool generations per push)
    wl.dl.set_option("max_chunk", 9_000_000)
    wl.step()
    cc, tc, mc = wl.result()
    assert mc == matched and np.array_equal(cc, counts)
    wl.dl.set_option("max_chunk", 1 << 27)
    # oracle on the first 1.5M reads
    m = 1_500_000
    wl.step(0, m)
    cp, tp, mp = wl.result()
    lib_text = synth.library_fasta(wl.lib_seqs)
    lib = O.Library(lib_text)
    ctr = O.Counter(lib, O.Permuter(lib), False, 30, 20, True)
    for first in range(0, m, 500_000):
        ctr.feed_text(synth.fastq_host(wl.lib_seqs, first, 500_000))
    assert (cp.tolist(), tp, mp) == (ctr.table(), ctr.total_reads(), ctr.matched_reads())
    wl.close()


@pytest.mark.parametrize("exact,mode", [(True, 0), (False, 1), (True, 1)])
def test_full_size_other_configs(env, exact, mode, tmp_path):
    """BASELINE.json configs[1] (-x) and configs[3] (variable adapter, auto offset) at their full size — 100k guides,
    100M reads: totals, count-sum invariant, linearity over a split, agreement of the generic kernels (variant 1) with
    the shipped path, the offsetter's answer on the sample's own text (STAGGER: Forward(30), offsetter.rs:185-210),
    and the oracle on a 1.5M-read prefix."""
    import os
    torch, S, synth, workload = env
    ffi = S._ffi
    from sgcount_amd import hostlib
    n, ng = 100_000_000, 100_000
    wl = workload.DeviceWorkload(n, ng, 20, one_mismatch=not exact, mode=mode)
    lib_text = synth.library_fasta(wl.lib_seqs)
    if mode == synth.MODE_STAGGER:
        lp, rp = os.path.join(str(tmp_path), "lib.fa"), os.path.join(str(tmp_path), "head.fq")
        open(lp, "wb").write(lib_text)
        open(rp, "wb").write(synth.fastq_host(wl.lib_seqs, 0, 6000, mode=mode))
        assert hostlib.entropy_offset_group(lp, [rp]) == [(False, 30)]
    wl.step()
    counts, total, matched = wl.result()
    assert total == n and int(counts.sum()) == matched
    # FIXED: 85 % exact + 2 + 2 % shifted (+ 5 + 1 % with one mismatch); STAGGER loses the 28/32-base prefixes (10 %)
    lo, hi = {(True, 0): (0.885, 0.895), (False, 1): (0.84, 0.87), (True, 1): (0.78, 0.82)}[(exact, mode)]
    assert lo < matched / total < hi, matched / total
    ffi.check(wl.abi.sgc_sample_reset(wl.sample))
    cut = 61_000_003
    for first, m in ((0, cut), (cut, n - cut)):
        ffi.check(wl.abi.sgc_sample_push_packed(wl.sample, wl.records.data_ptr() + 8 * first, m, ffi.MEM_DEVICE))
    ffi.check(wl.abi.sgc_sample_export_device(wl.sample, wl.export.data_ptr()))
    c2, t2, m2 = wl.result()
    assert t2 == total and m2 == matched and np.array_equal(c2, counts)
    wl.dl.set_option("variant", 1)
    wl.step()
    c1, t1, m1 = wl.result()
    assert t1 == total and m1 == matched and np.array_equal(c1, counts)
    wl.dl.set_option("variant", 4)
    m = 1_500_000
    wl.step(0, m)
    cp, tp, mp = wl.result()
    lib = O.Library(lib_text)
    ctr = O.Counter(lib, None if exact else O.Permuter(lib), False, 30, 20, True)
    for first in range(0, m, 500_000):
        ctr.feed_text(synth.fastq_host(wl.lib_seqs, first, 500_000, mode=mode))
    assert (cp.tolist(), tp, mp) == (ctr.table(), ctr.total_reads(), ctr.matched_reads())
    wl.close()


def test_more_than_2_32_reads_on_one_guide(env):
    """Counter::count tallies in usize (src/counter.rs:18); the device counts in u32 per guide and folds into u64 before any
    counter could wrap: 45 pushes of 100M copies of ONE read put 4.5e9 > 2^32 on a single guide (and on one slice, one slot)."""
    import ctypes as C
    torch, S, synth, workload = env
    ffi = S._ffi
    lib_seqs, library = workload.synth_library(100_000, 20)
    read = b"T" * 30 + lib_seqs[31337].tobytes() + b"G" * 40
    rec = S.pack_reads_host([read], 20, S.Offset.Forward(30), True)
    n, reps = 100_000_000, 45
    big = torch.full((n,), int(rec.view(np.int64)[0]), dtype=torch.int64, device="cuda")
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    smp = C.c_void_p()
    ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, 30, 1))
    for _ in range(reps):
        ffi.check(dl.lib.sgc_sample_push_packed(smp, big.data_ptr(), n, ffi.MEM_DEVICE))
    out = np.zeros(100_000, dtype=np.uint64)
    t, m = C.c_uint64(), C.c_uint64()
    ffi.check(dl.lib.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m)))
    dl.lib.sgc_sample_free(smp)
    assert t.value == m.value == reps * n > 2 ** 32
    assert int(out[31337]) == reps * n and int(out.sum()) == reps * n


def test_extreme_skew(env):
    """Adversarial input: 40M reads of which 30M are ONE guide (one slice, one slot), 6M a single one-mismatch
    variant of another guide, 4M one junk read.  Exercises K1's multi-block spills, K2's list overflow/rescan
    path and same-address LDS atomics; the table must still be exact."""
    import ctypes as C
    torch, S, synth, workload = env
    ffi = S._ffi
    lib_seqs, library = workload.synth_library(100_000, 20)
    g0, g1 = lib_seqs[4242].tobytes(), bytearray(lib_seqs[77].tobytes())
    g1[7] = ord("A") if g1[7] != ord("A") else ord("C")
    pre, tail = b"T" * 30, b"G" * 40
    reads = [pre + g0 + tail, pre + bytes(g1) + tail, pre + b"ACGT" * 5 + tail]
    mult = [30_000_000, 6_000_000, 4_000_000]
    lib_text = synth.library_fasta(lib_seqs)
    olib = O.Library(lib_text)
    operm = O.Permuter(olib)
    want = np.zeros(100_000, dtype=np.uint64)
    matched = 0
    for r, m in zip(reads, mult):
        t = O.Counter(olib, operm, False, 30, 20, True).feed_seq(r).table()
        want += np.array(t, dtype=np.uint64) * np.uint64(m)
        matched += sum(t) * m
    assert want[4242] == 30_000_000 and matched >= 30_000_000
    recs = S.pack_reads_host(reads, 20, S.Offset.Forward(30), True)
    d = torch.from_numpy(recs.view(np.int64)).cuda()
    big = torch.repeat_interleave(d, torch.tensor(mult, device="cuda"))
    big = big[torch.randperm(big.numel(), device="cuda")]          # interleave the three kinds
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    for variant, balanced in ((4, 1), (4, 0), (3, 1)):
        dl.set_option("variant", variant)
        dl.set_option("balanced", balanced)
        smp = C.c_void_p()
        ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, 30, 1))
        ffi.check(dl.lib.sgc_sample_push_packed(smp, big.data_ptr(), big.numel(), ffi.MEM_DEVICE))
        out = np.zeros(100_000, dtype=np.uint64)
        t, m = C.c_uint64(), C.c_uint64()
        ffi.check(dl.lib.sgc_sample_finish(smp, out.ctypes.data, C.byref(t), C.byref(m)))
        dl.lib.sgc_sample_free(smp)
        assert t.value == sum(mult) and m.value == matched, variant
        assert np.array_equal(out, want), variant
