"""CPU-side checks of the C ABI: the library loads, exports every declared symbol, and the host
packer agrees with a pure-Python restatement.  No compute entry point is called without a GPU."""
import os
import random
import re

import numpy as np
import pytest

import _records as R
from conftest import ROOT


def _ffi():
    from sgcount_amd import _ffi
    return _ffi


def test_library_loads_and_exports_every_declared_symbol():
    ffi = _ffi()
    lib = ffi.load()
    header = open(os.path.join(ROOT, "include", "sgcount_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sgc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(ffi.SYMBOLS), (declared ^ set(ffi.SYMBOLS))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sgc_version().startswith(b"sgcount_hip")


def test_record_bytes():
    lib = _ffi().load()
    assert [lib.sgc_record_bytes(L) for L in (0, 1, 20, 23, 24, 30, 31)] == [0, 8, 8, 8, 16, 16, 0]


def test_init_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ctypes as C
    ffi = _ffi()
    ctx = C.c_void_p()
    rc = ffi.load().sgc_init(0, C.byref(ctx))
    assert rc == ffi.E_HIP and not ctx.value
    assert b"no CPU fallback" in ffi.load().sgc_last_error()
    import sgcount_amd as S
    lib = S.Library.from_reader(S.parse_fastx(b">a\nACGT\n"))
    with pytest.raises(ffi.SgcError):
        S.Counter.new(S.parse_fastx(b">r\nACGT\n"), lib, None, S.Offset.Forward(0), 4, False)


@pytest.mark.parametrize("L", [1, 4, 6, 20, 23, 24, 30])
@pytest.mark.parametrize("reverse", [False, True])
def test_host_packer_matches_python_restatement(L, reverse):
    import sgcount_amd as S
    rng = random.Random(L * 2 + reverse)
    alpha = b"ACGT" * 6 + b"NnJ" + b"R"
    for o in (0, 1, 3):
        for recursion in (True, False):
            reads = []
            for _ in range(300):
                n = rng.choice([0, 1, L - 1, L, L + 1, o + L - 1, o + L, o + L + 1, o + L + 2, o + L + 7])
                reads.append(bytes(rng.choice(alpha) for _ in range(max(n, 0))))
            off = S.Offset.Reverse(o) if reverse else S.Offset.Forward(o)
            recs = S.pack_reads_host(reads, L, off, recursion)
            wpr = len(recs) // len(reads)
            assert wpr == (1 if L <= 23 else 2)
            for i, r in enumerate(reads):
                words = tuple(int(x) for x in recs[i * wpr:(i + 1) * wpr])
                assert R.decode_record(words, L) == R.expected_windows(r, L, reverse, o, recursion), (r, o, recursion)


def test_host_packer_rejects_unsupported_length():
    import sgcount_amd as S
    with pytest.raises(_ffi().SgcError):
        S.pack_reads_host([b"A" * 40], 31, S.Offset.Forward(0), True)


def test_k2_group_hash_is_balanced():
    """sgc_part.hip K2: workgroup g of a slice takes the blocks whose id b has ((b * 0x9E3779B1) mod 2^32 >> 16) & (G-1)
    == g.  Any assignment is correct (an overfull list is rescanned); this only checks the spread is even, so that
    the G workgroups of a slice get equal shares."""
    for G in (8, 16, 64, 512):
        ids = np.arange(0, 1 << 18, dtype=np.uint64)
        h = (((ids * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(16)) & np.uint64(G - 1)
        counts = np.bincount(h.astype(np.int64), minlength=G)
        assert counts.max() < 1.1 * counts.mean() + 8 and counts.min() > 0.9 * counts.mean() - 8, (G, counts.min(), counts.max())


def test_core_mix_is_a_balanced_bijection():
    """sgc_format.h sgc_core_mix: (core value * 0x9E3779B1) mod 2^(2 cl) must be a bijection — the five-byte slice records of
    k_partition drop the slice index and k_count_slices rebuilds the core value with the inverse multiplier 0x0E8B2F51 — and
    its top bits, which pick the library slice and the partition of core pass A, must spread both all core values and random
    subsets of them (a library's guides) evenly."""
    M, MINV = 0x9E3779B1, 0x0E8B2F51
    assert (M * MINV) % (1 << 32) == 1
    rng = np.random.default_rng(3)
    for cl in (1, 4, 9, 11, 14):
        bits = 2 * cl
        mask = (1 << bits) - 1
        n = min(1 << bits, 1 << 20)
        x = np.arange(1 << bits, dtype=np.uint64) if n == 1 << bits else rng.integers(0, 1 << bits, n, dtype=np.uint64)
        h = (x * np.uint64(M)) & np.uint64(mask)
        assert np.array_equal((h * np.uint64(MINV)) & np.uint64(mask), x)
        if n == 1 << bits:
            assert len(np.unique(h)) == n
        if bits >= 16:
            sample = rng.choice(x, 100_000, replace=False) if n > 100_000 else x
            hs = (sample * np.uint64(M)) & np.uint64(mask)
            for lp in (6, 8):
                counts = np.bincount((hs >> np.uint64(bits - lp)).astype(np.int64), minlength=1 << lp)
                assert counts.max() < 1.25 * counts.mean() + 8 and counts.min() > 0.75 * counts.mean() - 8, (cl, lp, counts.min(), counts.max())


def test_status_division_magic():
    """k_core decodes status = sC + K (sP + K sM) with n // K == (n * (2^20 // K + 1)) >> 20; exact for every
    status a record can carry (n < K^3, K = L + 2 <= 25 for one-u64 records)."""
    for K in range(3, 26):
        m = (1 << 20) // K + 1
        assert all(((n * m) >> 20) == n // K for n in range(K ** 3)), K


@pytest.mark.parametrize("n,L,alpha", [(100_000, 20, b"ACGT"), (5_000, 23, b"ACGT"), (300, 12, b"ACGT"), (2, 4, b"ACGT"),
                                      (40_000, 20, b"ACGTN"), (3_000, 34, b"ACGT"), (5, 1, b"ACGTN")])
def test_host_tables_reach_every_guide(n, L, alpha):
    """sgc_check_host_tables: the tables sgc_set_library would upload, built on the host and probed the way the kernels probe
    them — the open-addressed array (home bucket, wrap inside the slice), the one-slot two-choice image that k_count_slices
    stages in LDS (home slot or its alternate), and for libraries outside ACGT / longer than 30 the byte-string tables."""
    import ctypes as C
    ffi = _ffi()
    lib = ffi.load()
    rng = np.random.default_rng(n + L)
    seqs = set()
    while len(seqs) < n:
        need = n - len(seqs)
        draw = rng.integers(0, len(alpha), size=(need + 16, L))
        for row in np.frombuffer(alpha, dtype=np.uint8)[draw]:
            seqs.add(row.tobytes())
            if len(seqs) == n:
                break
    flat = b"".join(sorted(seqs))
    stats = (C.c_uint64 * 4)()
    rc = lib.sgc_check_host_tables(flat, n, L, 1 if n <= 5000 else 0, stats)
    assert rc == 0, lib.sgc_last_error()
    packed = set(alpha) <= set(b"ACGT") and L <= 30
    assert stats[0] == (1 if packed else 2)
    if packed:
        assert stats[1] >= 2 * n and stats[3] == 1          # load <= 0.5; the two-choice image was placed
    elif n <= 5000:
        assert stats[2] <= n * L * 4 and (stats[2] > 0 or L == 1)   # children stored (all five 1-mers: every child is a guide)
    # a duplicate sequence is reported as such by both builders
    dup = flat[:L] * 2
    assert lib.sgc_check_host_tables(dup, 2, L, 0, stats) == ffi.E_DUPLICATE
