"""Synthetic workload generator (host side): determinism, shape of the class mix, library properties."""
import collections
import os
import re

import numpy as np

from conftest import ROOT


def _syn():
    from sgcount_amd import synth
    synth.load()
    return synth


def test_synth_exports_every_declared_symbol():
    syn = _syn()
    header = open(os.path.join(ROOT, "include", "sgcount_synth.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sgs_[a-z0-9_]+)\s*\(", header))
    assert declared == set(syn.SYMBOLS), declared ^ set(syn.SYMBOLS)


def test_library_distinct_and_planted_pairs():
    syn = _syn()
    lib = syn.library(5000, 20)
    rows = [r.tobytes() for r in lib]
    assert len(set(rows)) == 5000 and set(b"".join(rows)) <= set(b"ACGT")
    assert np.array_equal(lib, syn.library(5000, 20))               # deterministic
    tail = rows[-200:]
    for p in range(100):
        a, b = tail[2 * p], tail[2 * p + 1]
        assert sum(x != y for x, y in zip(a, b)) == (2 if p & 1 else 1)
    fa = syn.library_fasta(lib)
    assert fa.startswith(b">sg000000\n" + rows[0] + b"\n>sg000001\n") and fa.count(b">") == 5000


def test_reads_slices_and_fastq_agree():
    syn = _syn()
    lib = syn.library(1000, 20)
    seqs, offs = syn.reads_host(lib, 0, 3000)
    s2, o2 = syn.reads_host(lib, 1000, 500)
    a, b = int(offs[1000]), int(offs[1500])
    assert np.array_equal(seqs[a:b], s2) and np.array_equal(offs[1000:1501] - offs[1000], o2)
    fq = syn.fastq_host(lib, 0, 3000).split(b"\n")
    assert fq[-1] == b"" and len(fq) == 4 * 3000 + 1
    for i in (0, 1, 999, 2999):
        assert fq[4 * i] == b"@r%d" % i
        assert fq[4 * i + 1] == seqs[int(offs[i]):int(offs[i + 1])].tobytes()
        assert fq[4 * i + 2] == b"+" and fq[4 * i + 3] == b"I" * int(offs[i + 1] - offs[i])


def test_class_mix_and_layout():
    syn = _syn()
    lib = syn.library(1000, 20)
    n = 40000
    seqs, offs = syn.reads_host(lib, 0, n)
    cls = collections.Counter()
    hot = 0
    for i in range(n):
        c, g = syn.read_class(i, 1000)
        cls[c] += 1
        hot += (g % 100 == 0) and c != 5
        r = seqs[int(offs[i]):int(offs[i + 1])].tobytes()
        guide = lib[g].tobytes()
        if c == 0:
            assert len(r) == 150 and r[30:50] == guide and r[:30] == b"TCTTGTGGAAAGGACGAAACACCGGTACCG"
        elif c == 1:
            assert sum(x != y for x, y in zip(r[30:50], guide)) == 1 and b"N" not in r
        elif c == 2:
            assert r[30:50].count(b"N") == 1
        elif c == 3:
            assert r[31:51] == guide and len(r) == 150
        elif c == 4:
            assert r[29:49] == guide and len(r) == 150
        elif c == 6:
            assert len(r) == 40
    frac = {k: v / n for k, v in cls.items()}
    for k, want in {0: .85, 1: .05, 2: .01, 3: .02, 4: .02, 5: .04, 6: .01}.items():
        assert abs(frac[k] - want) < 0.006, (k, frac[k])
    # 10 hot guides x 50 / (10*50 + 990) = 33.6 % of guide draws
    assert abs(hot / (n - cls[5]) - 500 / 1490) < 0.02


def test_stagger_mode_prefix_lengths():
    syn = _syn()
    lib = syn.library(500, 20)
    seqs, offs = syn.reads_host(lib, 0, 20000, mode=syn.MODE_STAGGER)
    starts = collections.Counter()
    for i in range(20000):
        c, g = syn.read_class(i, 500, mode=syn.MODE_STAGGER)
        if c != 0:
            continue
        r = seqs[int(offs[i]):int(offs[i + 1])].tobytes()
        starts[r.find(lib[g].tobytes(), 25)] += 1
    tot = sum(starts.values())
    for p, want in {28: .05, 29: .10, 30: .70, 31: .10, 32: .05}.items():
        assert abs(starts[p] / tot - want) < 0.015, (p, starts[p] / tot)
