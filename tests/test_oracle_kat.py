"""Pins the CPU oracle against the reference's own known-answer tests.

Each test re-expresses one #[test] of the reference (file:line in the
docstring) against oracle/sgcount_oracle.c.  No GPU needed.
"""
import math

import numpy as np
import pytest

import _oracle as O

LIB_ACTG = b">seq.0\nACTG\n"


def _counter(read, permute):
    lib = O.Library(LIB_ACTG)
    perm = O.Permuter(lib) if permute else None
    return O.Counter(lib, perm, False, 0, 4, False).feed_text(b">seq.0\n" + read + b"\n")


# ---- counter.rs ----
def test_count_no_distance_no_permute():
    """counter.rs:283-288"""
    assert _counter(b"ACTG", False).get_value(b"seq.0") == 1


def test_count_no_distance_with_permute():
    """counter.rs:291-296 (reader AGTG, permuter None)"""
    assert _counter(b"AGTG", False).get_value(b"seq.0") == 0


def test_count_with_distance_no_permute():
    """counter.rs:299-304"""
    assert _counter(b"AGTG", False).get_value(b"seq.0") == 0


def test_count_with_distance_with_permute():
    """counter.rs:307-320"""
    c = _counter(b"AGTG", True)
    assert c.get_value(b"seq.0") == 1
    assert c.total_reads() == 1 and c.matched_reads() == 1


@pytest.mark.parametrize("pos,expect", [(O.POS_NULL, (4, 8)), (O.POS_PLUS, (5, 9)), (O.POS_MINUS, (3, 7))])
def test_bounds_checking(pos, expect):
    """counter.rs:323-352: bounds(b"ACTGACTGACTG", 4, 4, pos)"""
    assert O.bounds(12, 4, 4, pos) == expect


@pytest.mark.parametrize("pos,length", [(O.POS_NULL, 7), (O.POS_PLUS, 7), (O.POS_MINUS, 6)])
def test_bounds_checking_clipped(pos, length):
    """counter.rs:355-382"""
    assert O.bounds(length, 4, 4, pos) is None


def test_bounds_edges():
    """counter.rs:166-179: Minus at offset 0 is None; max == len is allowed"""
    assert O.bounds(100, 0, 4, O.POS_MINUS) is None
    assert O.bounds(8, 4, 4, O.POS_NULL) == (4, 8)
    assert O.bounds(8, 4, 4, O.POS_CENTERED) == (4, 8)


# ---- permutes.rs ----
def test_permuter_validate_singleton():
    """permutes.rs:193-207"""
    p = O.Permuter(seqs=[b"ACTG"])
    truth = [b"AATG", b"ACGG", b"ACAG", b"TCTG", b"ACNG", b"NCTG", b"ACTA", b"GCTG", b"AGTG", b"ACTC", b"ATTG",
             b"ANTG", b"ACCG", b"ACTT", b"CCTG", b"ACTN"]
    assert all(p.contains(t) == b"ACTG" for t in truth)
    assert not any(p.null_contains(t) for t in truth)
    assert p.null_contains(b"ACTG")
    assert p.null_len() == 1
    assert p.map_len() == 16


def test_permuter_validate_positive():
    """permutes.rs:210-231"""
    p = O.Permuter(seqs=[b"AC", b"CG"])
    pos = [b"GC", b"TC", b"NC", b"AA", b"AT", b"AN", b"CA", b"CT", b"CN", b"GG", b"TG", b"NG"]
    assert all(p.contains(t) is not None for t in pos)
    assert p.map_len() == 12
    assert not any(p.null_contains(t) for t in pos)
    for t in pos[:6]:
        assert p.contains(t) == b"AC"
    for t in pos[6:]:
        assert p.contains(t) == b"CG"


def test_permuter_validate_negative():
    """permutes.rs:234-253"""
    p = O.Permuter(seqs=[b"AC", b"CG"])
    neg = [b"AG", b"CG", b"CC", b"AG"]
    assert all(p.null_contains(t) for t in neg)
    assert p.null_len() == 4
    assert all(p.contains(t) is None for t in neg)


def test_permuter_order_independent():
    """permutes.rs:63-75 iterates HashMap order; observable lookups must not depend on it"""
    seqs = [b"ACGT", b"ACGA", b"TTTT", b"TTTA", b"CCCC", b"GGGG", b"ACCA"]
    import itertools
    import random
    rng = random.Random(7)
    base = None
    libset = set(seqs)
    for _ in range(6):
        s = seqs[:]
        rng.shuffle(s)
        p = O.Permuter(seqs=s)
        res = {}
        for q in itertools.product(b"ACGTN", repeat=4):
            q = bytes(q)
            if q in libset:
                continue  # assign() consults the library first (counter.rs:111)
            res[q] = p.contains(q)
        if base is None:
            base = res
        # children equal to the unique-parent rule
        for q, par in res.items():
            near = [g for g in seqs if sum(a != b for a, b in zip(g, q)) == 1]
            assert par == (near[0] if len(near) == 1 else None), (q, par, near)
        assert res == base


# ---- library.rs ----
def test_library_build():
    """library.rs:119-123"""
    lib = O.Library(LIB_ACTG)
    assert lib.size() == 4 and lib.n() == 1


def test_library_validate_contains():
    """library.rs:126-130"""
    lib = O.Library(LIB_ACTG)
    assert lib.contains(b"ACTG") == b"seq.0"
    assert lib.contains(b"ACTT") is None


def test_library_duplicates():
    """library.rs:133-136 should_panic"""
    with pytest.raises(O.OracleError) as e:
        O.Library(b">seq.0\nACTG\n>seq.1\nACTG\n")
    assert e.value.code == O.E_DUPLICATE_SEQ


def test_library_inconsistent_and_empty():
    """library.rs:79-85 error; :74 unwrap on empty"""
    with pytest.raises(O.OracleError) as e:
        O.Library(b">a\nACTG\n>b\nACT\n")
    assert e.value.code == O.E_INCONSISTENT
    with pytest.raises(O.OracleError) as e:
        O.Library(b"")
    assert e.value.code == O.E_EMPTY


# ---- offsetter.rs ----
READER = b">seq.0\nACT\n>seq.1\nACC\n>seq.2\nACT\n"
READER_N = b">seq.0\nACT\n>seq.1\nACC\n>seq.2\nACT\n>seq.3\nACN\n"
OFFSET_READER = b">seq.0\nAACAAACT\n>seq.1\nAACAAACC\n>seq.2\nAACAAACT\n"
RC_OFFSET_READER = b">seq.0\nAGTTTGTT\n>seq.1\nGGTTTGTT\n>seq.2\nAGTTTGTT\n"


def test_offsetter_minimization():
    """offsetter.rs:249-256"""
    rev, idx = O.minimize_mse(list(np.linspace(0., 10., 11)), list(np.linspace(10., 20., 100)))
    assert (rev, idx) == (False, 0)


def test_offsetter_undersized_minimization():
    """offsetter.rs:259-263"""
    with pytest.raises(O.OracleError) as e:
        O.minimize_mse(list(np.linspace(0., 10., 11)), list(np.linspace(10., 20., 5)))
    assert e.value.code == O.E_SHORT


def test_offsetter_positional_counts():
    """offsetter.rs:266-283 (size 3; first record consumed, two counted)"""
    m = O.position_counts(READER)
    assert m == [[2.0, 0.0, 0.0, 0.0], [0.0, 2.0, 0.0, 0.0], [0.0, 1.0, 0.0, 1.0]]


def test_offsetter_position_counts_with_n():
    """offsetter.rs:353-362.  The upstream assertion is only `(posmat - expected).sum() == 0`
    with expected row 2 = [2,1,2,1]; the code (offsetter.rs:65-76) yields [1,2,1,2] for ACC,ACT,ACN
    (same sum), so assert upstream's check literally and the exact matrix separately."""
    m = np.array(O.position_counts(READER_N))
    expected = np.array([[3.0, 0.0, 0.0, 0.0], [0.0, 3.0, 0.0, 0.0], [2.0, 1.0, 2.0, 1.0]])
    assert (m - expected).sum() == 0.0
    assert m.tolist() == [[3.0, 0.0, 0.0, 0.0], [0.0, 3.0, 0.0, 0.0], [1.0, 2.0, 1.0, 2.0]]


def test_offsetter_entropy_values():
    """offsetter.rs:286-300 normalize + :90-95 entropy: rows [1,0,0,0],[0,1,0,0],[0,.5,0,.5]"""
    h = O.positional_entropy(READER)
    assert h[0] == 0.0 and h[1] == 0.0
    assert math.isclose(h[2], math.log(2.0), rel_tol=0, abs_tol=1e-15)


def test_offsetter_offset():
    """offsetter.rs:303-315 ⇒ Forward(5)"""
    assert O.entropy_offset(READER, OFFSET_READER) == (False, 5)


def test_offsetter_rc_offset():
    """offsetter.rs:318-328 ⇒ Reverse(5)"""
    assert O.entropy_offset(READER, RC_OFFSET_READER) == (True, 5)


def test_offsetter_subsample_take():
    """main.rs:117 / offsetter.rs:172-173 .take(subsample): the size-probe record counts toward take"""
    assert O.position_counts(READER_N, take=3) == [[2.0, 0, 0, 0], [0, 2.0, 0, 0], [0, 1.0, 0, 1.0]]
    with pytest.raises(O.OracleError):
        O.position_counts(READER_N, take=0)


# ---- results.rs ----
RES_LIB = b">sgrna1\nACTG\n>sgrna2\nGTCA\n"
RES_GENEMAP = b"GENE1\tsgrna1\nGENE2\tsgrna2\n"


def test_results_generate_columns():
    """results.rs:134-139"""
    assert O.generate_columns(["A", "B"]) == "Guide\tA\tB"
    assert O.generate_columns(["A", "B"], with_genemap=True) == "Guide\tGene\tA\tB"


def test_results_append_count_and_gene():
    """results.rs:164-177: a count cell is "\t100", a gene cell "\tGENE1"; rows = alias[\tgene]\tc1\tc2"""
    lib = O.Library(RES_LIB)
    gm = O.GeneMap(RES_GENEMAP)
    text = O.format_results(lib, [[100, 200], [100, 200]], ["sample1", "sample2"], gm, include_zero=True)
    assert text == "Guide\tGene\tsample1\tsample2\nsgrna1\tGENE1\t100\t100\nsgrna2\tGENE2\t200\t200\n"
    text = O.format_results(lib, [[100, 0], [0, 0]], ["s1", "s2"], None, include_zero=False)    # results.rs:90-94
    assert text == "Guide\ts1\ts2\nsgrna1\t100\t0\n"
    text = O.format_results(lib, [[100, 0], [0, 0]], ["s1", "s2"], None, include_zero=True)
    assert text == "Guide\ts1\ts2\nsgrna1\t100\t0\nsgrna2\t0\t0\n"


def test_results_missing_gene_panics():
    """results.rs:59"""
    lib = O.Library(RES_LIB)
    with pytest.raises(O.OracleError) as e:
        O.format_results(lib, [[1, 1]], ["s"], O.GeneMap(b"GENE1\tsgrna1\n"), True)
    assert e.value.code == O.E_NOGENE


# ---- genemap.rs ----
GM_TEXT = b"gene1\tsgrna1\ngene2\tsgrna2\ngene3\tsgrna3\n"


def test_genemap_build():
    """genemap.rs:124-130"""
    g = O.GeneMap(GM_TEXT)
    assert (g.get(b"sgrna1"), g.get(b"sgrna2"), g.get(b"sgrna3")) == (b"gene1", b"gene2", b"gene3")
    assert g.get(b"sgrna4") is None


def test_genemap_validate_library():
    """genemap.rs:133-148 (library sequences may be any bytes, e.g. lowercase)"""
    g = O.GeneMap(GM_TEXT)
    assert g.missing_aliases(O.Library(b">sgrna1\nACTG\n>sgrna2\ngtca\n>sgrna3\nTCAG\n")) is None
    assert g.missing_aliases(O.Library(b">sgrna1\nACTG\n>sgrna4\ngtca\n")) == b"sgrna4"


def test_genemap_from_file():
    """genemap.rs:151-156 example/g2s.txt"""
    import os
    from conftest import DATA
    g = O.GeneMap(open(os.path.join(DATA, "g2s.txt"), "rb").read())
    assert g.get(b"lib.0") == b"gene.0" and g.get(b"lib.99") == b"gene.9"


def test_genemap_errors():
    """genemap.rs:58 missing tab; :60-64 duplicate sgRNA"""
    with pytest.raises(O.OracleError) as e:
        O.GeneMap(b"gene1 sgrna1\n")
    assert e.value.code == O.E_NOTAB
    with pytest.raises(O.OracleError) as e:
        O.GeneMap(b"gene1\tsgrna1\ngene2\tsgrna1\n")
    assert e.value.code == O.E_DUPKEY


# ---- utils.rs ----
def test_utils_sample_names():
    """utils.rs:55-81"""
    paths = ["example/some_name_1.fastq.gz", "example/some_name_2.fastq", "example/some_name_3.fasta.gz",
             "example/some_name_4.fasta", "example/some_name_5.fq.gz", "example/some_name_6.fq",
             "example/some_name_7.fa.gz", "example/some_name_8.fa"]
    names, fb = O.generate_sample_names(paths)
    assert names == ["some_name_%d" % i for i in range(1, 9)] and not fb


def test_utils_sample_names_duplicates():
    """utils.rs:84-104"""
    paths = ["example/some_name_1.fastq.gz", "example/some_name_1.fastq", "example/some_name_3.fasta.gz",
             "example/some_name_4.fasta", "example/some_name_5.fq.gz", "example/some_name_6.fq",
             "example/some_name_7.fa.gz", "example/some_name_8.fa"]
    names, fb = O.generate_sample_names(paths)
    assert names == ["Sample.%d" % i for i in range(8)] and fb
