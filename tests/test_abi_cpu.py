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


def test_status_division_magic():
    """k_core decodes status = sC + K (sP + K sM) with n // K == (n * (2^20 // K + 1)) >> 20; exact for every
    status a record can carry (n < K^3, K = L + 2 <= 25 for one-u64 records)."""
    for K in range(3, 26):
        m = (1 << 20) // K + 1
        assert all(((n * m) >> 20) == n // K for n in range(K ** 3)), K
