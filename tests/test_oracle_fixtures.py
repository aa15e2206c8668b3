"""Oracle vs the reference's example/ data files and the committed golden vectors (CPU only)."""
import collections
import hashlib
import json
import os

import pytest

import _oracle as O
from conftest import GOLDEN

# SURVEY.md §8c expectations (derived by hand-restating counter.rs; reproduced here by the C oracle)
EXPECT = {
    "sequence": (1000, 1000, 100, "9659a9ad5c1c5d2f"),
    "zero.sequence": (895, 895, 90, "d089981e96942016"),
    "diff.sequence": (1101, 1000, 100, "9659a9ad5c1c5d2f"),
    "offset": (1000, 1000, 100, "9659a9ad5c1c5d2f"),
    "offset_clipped": (1000, 999, 100, "97f3b1c6408c5215"),
}


def _sha(ids, counts):
    rows = sorted(f"{i.decode()}\t{c}" for i, c in zip(ids, counts) if c > 0)
    return hashlib.sha256("\n".join(rows).encode()).hexdigest()[:16]


@pytest.mark.parametrize("name", list(EXPECT))
@pytest.mark.parametrize("exact", [True, False])
def test_example_fixture_tables(name, exact, example_library_text, example_reads):
    total, matched, nhit, sha = EXPECT[name]
    counts, tot, mat = O.count_text(example_library_text, example_reads[name], False, 5, exact, True)
    ids = O.Library(example_library_text).ids()
    assert (tot, mat) == (total, matched)
    assert sum(1 for c in counts if c) == nhit
    assert sum(counts) == mat
    assert _sha(ids, counts) == sha


@pytest.mark.parametrize("name", ["sequence", "zero.sequence", "diff.sequence", "offset"])
def test_example_fixture_header_truth(name, example_library_text, example_reads):
    """The example reads are named '@seq.<GUIDE>.<n>': the generating guide is in the header, which
    gives a count table that does not depend on any restatement of counter.rs."""
    lib = O.Library(example_library_text)
    hdr = collections.Counter()
    lines = example_reads[name].split(b"\n")
    for i in range(0, len(lines) - 1, 4):
        if lines[i].startswith(b"@seq."):
            hdr[lines[i].split(b".")[1]] += 1
    truth = [hdr.get(s, 0) for s in lib.seqs()]
    counts, _, _ = O.count_text(example_library_text, example_reads[name], False, 5, True, True)
    assert counts == truth


def test_example_auto_offset(example_library_text, example_reads):
    """entropy auto-offset gives Forward(5) for all five example files"""
    for name, txt in example_reads.items():
        assert O.entropy_offset(example_library_text, txt) == (False, 5), name


def test_golden_example_counts_current(example_library_text, example_reads):
    g = json.load(open(os.path.join(GOLDEN, "example_counts.json")))
    for c in g["cases"]:
        counts, tot, mat = O.count_text(example_library_text, example_reads[c["file"][:-9]], False, g["offset"],
                                        c["exact"], c["position_recursion"])
        assert counts == c["counts"] and tot == c["total"] and mat == c["matched"]


def test_golden_edge_cases_current():
    """The committed edge-case vectors are what the oracle produces today (guards against drift)."""
    g = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))
    for c in g["cases"]:
        lib_text = b"".join(b">g%d\n%s\n" % (i, s.encode()) for i, s in enumerate(c["guides"]))
        lib = O.Library(lib_text)
        perm = None if c["exact"] else O.Permuter(lib)
        ctr = O.Counter(lib, perm, c["reverse"], c["offset"], lib.size(), c["position_recursion"])
        for r in c["reads"]:
            ctr.feed_seq(r.encode("latin1"))
        want = collections.Counter(a for a in c["assign"] if a >= 0)
        got = {i: v for i, v in enumerate(ctr.table()) if v}
        assert got == dict(want), c["name"]
        assert ctr.total_reads() == len(c["reads"])
        assert ctr.matched_reads() == sum(want.values())


def test_recursion_priority_and_bounds():
    """counter.rs:96-140: C-exact, C-1mm, P-exact, P-1mm, M-exact, M-1mm; bounds failure ends the chain"""
    lib = O.Library(b">a\nAAAAAA\n>b\nCAAAAA\n>c\nGGGGGG\n")
    perm = O.Permuter(lib)

    def one(read, recursion=True, permute=True, offset=2):
        c = O.Counter(lib, perm if permute else None, False, offset, 6, recursion)
        c.feed_seq(read)
        t = c.table()
        return t.index(1) if 1 in t else -1

    assert one(b"TTAAAAAATT") == 0
    # Centered window "TCAAAA" is not within 1 of anything; Plus "CAAAAA" is exact b
    assert one(b"TTTCAAAAATT") == 1
    # Centered 1mm (GGGGGT→c) wins over Plus exact
    assert one(b"TTGGGGGTCAAAAA") == 2
    # read ends exactly at Centered window and misses: Plus fails bounds ⇒ None, Minus never tried
    assert one(b"TGGGGGGT", permute=False) == -1
    assert one(b"TGGGGGGTT", permute=False) == 2   # now Plus is in bounds (miss), Minus hits
    assert one(b"TGGGGGGT") == 2                   # with the permuter, Centered GGGGGT is a 1mm of c
    assert one(b"GGGGGGTTT", offset=0) == 2        # Centered exact c
    assert one(b"TGGGGGGTT", recursion=False, permute=False) == -1
    # ambiguity: AAAAAA/CAAAAA differ at pos 0 ⇒ "GAAAAA","TAAAAA","NAAAAA" are ambiguous
    for r in (b"TTGAAAAATT", b"TTTAAAAATT", b"TTNAAAAATT"):
        assert one(r, recursion=False) == -1
    assert one(b"TTAAAAATTT", recursion=False) == 0     # AAAAAT is 1 from a, 2 from b
