#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference (Rust) cannot be built or run in this environment, so these
vectors come from oracle/sgcount_oracle.c, which is itself pinned by the
reference's own unit-test known answers (tests/test_oracle_kat.py) and by the
example/ FASTQ files whose read names carry the generating guide sequence
(an expectation that is independent of the oracle and asserted in
tests/test_oracle_fixtures.py).

Outputs
  example_counts.json   per example FASTQ x {exact, 1mm} x {recursion on/off}: counts (library order), total, matched
  edge_cases.json       small hand-made + seeded-random cases with per-read assignments
  fastq_endings.json    how a FASTQ stream may end (reader decision #3): the tail behind a body of whole records,
                        and what the count of the whole text must be — or that it must be refused
Run:  python tests/golden/make_golden.py
"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

DATA = os.path.join(os.path.dirname(HERE), "data")
NAMES = ["sequence", "zero.sequence", "diff.sequence", "offset", "offset_clipped"]


def example_counts():
    lib = gzip.open(os.path.join(DATA, "library.fasta.gz")).read()
    out = {"library": "library.fasta.gz", "offset": 5, "cases": []}
    for name in NAMES:
        txt = gzip.open(os.path.join(DATA, name + ".fastq.gz")).read()
        rev, idx = O.entropy_offset(lib, txt)
        for exact in (True, False):
            for recursion in (True, False):
                counts, tot, mat = O.count_text(lib, txt, False, 5, exact, recursion)
                out["cases"].append({"file": name + ".fastq.gz", "exact": exact, "position_recursion": recursion,
                                     "counts": counts, "total": tot, "matched": mat,
                                     "auto_offset": {"reverse": rev, "index": idx}})
    return out


def per_read(lib_text, reads, reverse, offset, exact, recursion):
    """Assignment (library index or -1) of each read, via one-read-at-a-time counters."""
    lib = O.Library(lib_text)
    perm = None if exact else O.Permuter(lib)
    ids = lib.ids()
    first_index = {}
    for i, ident in enumerate(ids):
        first_index.setdefault(ident, i)
    res = []
    for r in reads:
        c = O.Counter(lib, perm, reverse, offset, lib.size(), recursion)
        c.feed_seq(r)
        t = c.table()
        hit = [i for i, v in enumerate(t) if v]
        res.append(first_index[ids[hit[0]]] if hit else -1)
    return res


def fasta(seqs, prefix="g"):
    return b"".join(b">%s%d\n%s\n" % (prefix.encode(), i, s) for i, s in enumerate(seqs))


def mutate(rng, s, alphabet=b"ACGT"):
    i = rng.randrange(len(s))
    c = rng.choice([x for x in alphabet if x != s[i]])
    return s[:i] + bytes([c]) + s[i + 1:]


def rc(s):
    return bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(s))


def edge_cases():
    rng = random.Random(0x5EED)
    cases = []

    def add(name, guides, reads, offset, reverse=False):
        lib_text = fasta(guides)
        for exact in (False, True):
            for recursion in (True, False):
                cases.append({
                    "name": name, "guides": [g.decode() for g in guides], "reads": [r.decode("latin1") for r in reads],
                    "offset": offset, "reverse": reverse, "exact": exact, "position_recursion": recursion,
                    "assign": per_read(lib_text, reads, reverse, offset, exact, recursion)})

    # 1. hand-made: L=6, pairs at Hamming 1 (g0,g1) and Hamming 2 (g2,g3)
    G = [b"ACGTAC", b"ACGTAG", b"TTGCAA", b"TTGGTA", b"CCCCCC", b"GATTAC"]
    pre = b"TT"
    tail = b"GGGGGGGG"
    R = []
    for g in G:
        R.append(pre + g + tail)                 # exact at Centered
        R.append(pre + b"A" + g + tail)          # exact at Plus
        R.append(pre[:1] + g + tail)             # exact at Minus
    R += [pre + b"ACGTAA" + tail,               # 1mm of g0 AND g1 → ambiguous
          pre + b"ACGTAN" + tail,               # N child of g0 and g1 → ambiguous
          pre + b"ACNTAC" + tail,               # N child of g0 only
          pre + b"ANNTAC" + tail,               # two N
          pre + b"TTGCTA" + tail,               # between g2/g3 (Hamming 1 to both) → ambiguous
          pre + b"TTGCAT" + tail,               # 1mm g2 only
          pre + b"ccCCCC" + tail,               # lowercase ×2
          pre + b"CCCCCc" + tail,               # one lowercase byte: never a generated child
          pre + b"CCCCCR" + tail,               # IUPAC byte
          pre + b"CCCCC",                       # short: Centered out of bounds
          pre + b"CCCCCC",                      # len == offset+L: Plus fails bounds
          pre + b"CCCCCA",                      # 1mm, read ends at window
          pre + b"ACCCCCC",                     # Plus exact, read ends exactly
          b"",                                  # empty read
          b"T",
          pre + b"GATTAC" + b"GATTAC",          # repeats
          pre + b"CGATTAC" + tail,              # Plus exact; Centered is 1mm? (CGATTA vs GATTAC no)
          pre + b"CCCCCG" + b"ATTAC" + tail,    # Centered 1mm(g4) beats Plus...
          b"NN" + G[5] + tail,                  # N outside windows' core
          b"TN" + G[5] + tail,                  # N at Minus-only position
          pre + G[5] + b"N" + tail,             # N at Plus-only position
          pre + G[5][:5] + b"N" + tail]
    add("handmade_L6", G, R, 2)
    add("handmade_L6_offset0", G, [r[2:] for r in R], 0)
    # reverse: reads are reverse complements of the forward constructs, trailing pad so offset indexes rc string
    RR = [rc(r) for r in R if r]
    # literal N / lowercase in a reverse-strand read: complemented to a non-ACGTN byte ⇒ never matches
    for r in (pre + G[0] + tail, pre + G[4] + tail, pre + b"A" + G[5] + tail):
        q = bytearray(rc(r))
        q[len(q) - 1 - 4] = ord("N")
        RR.append(bytes(q))
        q = bytearray(rc(r))
        q[len(q) - 1 - 1] = ord("N")             # outside the Centered window
        RR.append(bytes(q))
    add("handmade_L6_reverse", G, RR, 2, reverse=True)

    # 2. seeded random, L=20, with planted near-duplicates
    def rand_seq(n, alpha=b"ACGT"):
        return bytes(rng.choice(alpha) for _ in range(n))
    guides = []
    seen = set()
    while len(guides) < 40:
        s = rand_seq(20)
        if s not in seen:
            seen.add(s); guides.append(s)
    for _ in range(6):                            # Hamming-1 and Hamming-2 neighbours
        s = mutate(rng, rng.choice(guides[:40]))
        if s not in seen:
            seen.add(s); guides.append(s)
        s2 = mutate(rng, mutate(rng, rng.choice(guides[:40])))
        if s2 not in seen:
            seen.add(s2); guides.append(s2)
    reads = []
    for k in range(400):
        g = rng.choice(guides)
        pre = rand_seq(7)
        cls = rng.random()
        if cls < 0.35:
            body = g
        elif cls < 0.5:
            body = mutate(rng, g)
        elif cls < 0.58:
            body = mutate(rng, g, b"ACGTN")
        elif cls < 0.66:
            body = mutate(rng, mutate(rng, g))
        elif cls < 0.74:
            pre = pre + rand_seq(1)
            body = g if rng.random() < 0.5 else mutate(rng, g)
        elif cls < 0.82:
            pre = pre[:-1]
            body = g if rng.random() < 0.5 else mutate(rng, g, b"ACGTN")
        elif cls < 0.9:
            body = rand_seq(20, b"ACGTN")
        else:
            body = g[: rng.randrange(0, 21)]
        r = pre + body + (rand_seq(rng.randrange(0, 6)) if cls < 0.9 else b"")
        if rng.random() < 0.05:
            i = rng.randrange(len(r)) if r else 0
            r = r[:i] + rng.choice([b"n", b"a", b"R", b"N"]) + r[i + 1:]
        reads.append(r)
    add("random_L20", guides, reads, 7)
    add("random_L20_reverse", guides, [rc(r) for r in reads], 7, reverse=True)

    # 3. maximum REC8 length (L=23) and a REC16 length (L=30)
    for L in (23, 24, 30):
        gs = []
        seen = set()
        while len(gs) < 12:
            s = rand_seq(L)
            if s not in seen:
                seen.add(s); gs.append(s)
        gs.append(mutate(rng, gs[0]))
        rs = []
        for k in range(80):
            g = rng.choice(gs)
            body = g if k % 3 == 0 else mutate(rng, g, b"ACGTN")
            pre = rand_seq(3 + (k % 5 == 1) - (k % 5 == 2))
            rs.append(pre + body + rand_seq(k % 4))
        add("random_L%d" % L, gs, rs, 3)
    return cases


FASTQ_ENDING_TAILS = [
    "@e\n\n+\n\n", "@e\n\n+\n", "@e\n\n+", "@e\n\n+\n\n\n\n", "@e\r\n\r\n+\r\n", "@e\r\n\r\n+\r\n\r\n",
    "@e\nTTGATTACGG\n+\n", "@e\nTTGATTACGG\n+", "@e\nTTGATTACGG\n+\nIIIIIIIIII", "\n\n\n", "",
    "@e\n\n", "@e\n", "@e", "@e\nTTGATTACGG\n", "@e\nTTGATTACGG\nIIIIIIIIII\n+\n",
]


def fastq_endings():
    """A body of whole records + a tail.  A stream that ends behind a separator line ends with a record whose quality line is empty;
    blank lines at the very end are not records; every other incomplete tail is refused."""
    rng = random.Random(20261005)
    guides = ["ACGTAC", "ACGTAG", "TTGCAA", "TTGGTA", "CCCCCC", "GATTAC"]
    lib_text = "".join(">g%d\n%s\n" % (i, g) for i, g in enumerate(guides)).encode()
    body = "".join("@r%d\nTT%sGG\n+\n%s\n" % (i, rng.choice(guides), "I" * 10) for i in range(200))
    out = {"library": lib_text.decode(), "offset": 2, "body": body, "cases": []}
    for tail in FASTQ_ENDING_TAILS:
        text = (body + tail).encode()
        try:
            counts, tot, mat = O.count_text(lib_text, text, False, 2, False, True)
            out["cases"].append({"tail": tail, "counts": counts, "total": tot, "matched": mat})
        except Exception:
            out["cases"].append({"tail": tail, "error": True})
    return out


def main():
    with open(os.path.join(HERE, "fastq_endings.json"), "w") as f:
        json.dump(fastq_endings(), f, separators=(",", ":"))
        f.write("\n")
    with open(os.path.join(HERE, "example_counts.json"), "w") as f:
        json.dump(example_counts(), f, separators=(",", ":"))
        f.write("\n")
    with open(os.path.join(HERE, "edge_cases.json"), "w") as f:
        json.dump({"cases": edge_cases()}, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
