#!/usr/bin/env python3
"""Randomised differential run on the GPU box: random libraries, reads, offsets, strands and modes through the C ABI (host-packed
records, device-packed reads, window pieces, FASTQ text) against the CPU oracle, for a number of seconds.

    python3 tools/fuzz_parity.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O            # noqa: E402
import sgcount_amd as S        # noqa: E402


def case(rng):
    L = rng.choice([4, 8, 12, 16, 19, 20, 20, 20, 21, 22, 23, 24, 27, 30])
    n_guides = rng.choice([1, 3, 50, 300, 2000, 6000])
    o = rng.choice([0, 1, 2, 5, 9, 30])
    alpha = b"ACGT"
    guides = sorted({bytes(rng.choice(alpha) for _ in range(L)) for _ in range(n_guides)})      # (sorted: a set's order changes from process to process)
    rng.shuffle(guides)
    for _ in range(min(len(guides) // 4, 200)):            # neighbours at distance 1 and 2
        g = bytearray(rng.choice(guides))
        for _ in range(rng.choice([1, 1, 2])):
            g[rng.randrange(L)] = rng.choice(alpha)
        if bytes(g) not in guides:
            guides.append(bytes(g))
    if rng.random() < 0.3 and len(guides) >= 4:              # a few (or many) guides outside ACGT: the hybrid / byte-string paths
        frac = rng.choice([0.02, 0.1, 0.7])
        for i in range(len(guides)):
            if rng.random() < frac:
                g = bytearray(guides[i]); g[rng.randrange(L)] = rng.choice(b"NNNna")
                if bytes(g) not in guides:
                    guides[i] = bytes(g)
    hot = rng.random() < 0.3
    reads = []
    for _ in range(rng.choice([1, 50, 3000, 20000])):
        g = bytearray(guides[0] if hot and rng.random() < 0.6 else rng.choice(guides))
        u = rng.random()
        if u < 0.25:
            g[rng.randrange(L)] = rng.choice(b"ACGTN")
        elif u < 0.3:
            g[rng.randrange(L)] = rng.choice(b"ACGTN"); g[rng.randrange(L)] = rng.choice(b"ACGTNa")
        elif u < 0.35:
            g = bytearray(rng.choice(b"ACGTN") for _ in range(L))
        pre = bytes(rng.choice(alpha) for _ in range(max(o + rng.choice([0, 0, 0, 1, -1, 2]), 0)))
        tail = bytes(rng.choice(alpha) for _ in range(rng.choice([0, 1, 2, 5, 30, 120])))
        r = pre + bytes(g) + tail
        if rng.random() < 0.03:
            r = r[: rng.randrange(len(r) + 1)]
        reads.append(r)
    return L, o, guides, reads


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    S._ffi.load()
    t0, n_cases, n_checks = time.time(), 0, 0
    last_note = time.time()
    while time.time() - t0 < seconds:
        if time.time() - last_note > 60:
            print("... %d cases so far" % n_cases, flush=True); last_note = time.time()
        L, o, guides, reads = case(rng)
        reverse = rng.random() < 0.4
        if reverse:
            reads = [bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r)) for r in reads]
        lib_text = b"".join(b">g%d\n%s\n" % (i, g) for i, g in enumerate(guides))
        reads_text = b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(reads))
        lib = S.Library.from_reader(S.parse_fastx(lib_text))
        perm = S.Permuter.new(lib.keys())
        off = S.Offset.Reverse(o) if reverse else S.Offset.Forward(o)
        exact, recursion = rng.random() < 0.3, rng.random() < 0.7
        only = int(os.environ.get("FUZZ_ONLY", "-1"))          # replay: draw everything, run only that case, say where it differs
        run = only < 0 or n_cases == only
        if run:
            want, tot, mat = O.count_text(lib_text, reads_text, reverse, o, exact, recursion)
        opts = rng.choice([{}, {}, {"balanced": 0}, {"five_byte": 0}, {"variant": 3}, {"direct": 0}])
        for pack in ("host", "device", "windows", "fastq"):
            if pack == "host" and L > 30:
                continue
            batch = rng.choice([1, 7, 4001, 1 << 20]) if len(reads) < 5000 else rng.choice([4001, 1 << 20])
            if not run:
                continue
            ctr = S.Counter.new(S.parse_fastx(reads_text), lib, None if exact else perm, off, L, recursion, pack=pack, batch=batch, options=opts)
            got = ctr.guide_counts().tolist()
            ok = got == want and (ctr.total_reads(), ctr.matched_reads()) == (tot, mat)
            n_checks += 1
            if not ok:
                print("MISMATCH seed %d case %d: L %d o %d reverse %s exact %s recursion %s pack %s batch %d opts %s guides %d reads %d" % (
                    seed, n_cases, L, o, reverse, exact, recursion, pack, batch, opts, len(guides), len(reads)))
                print("  totals got", (ctr.total_reads(), ctr.matched_reads()), "want", (tot, mat), " non-ACGT guides:",
                      sum(1 for g in guides if any(c not in b"ACGT" for c in g)))
                for i, (a, b) in enumerate(zip(got, want)):
                    if a != b:
                        print("  guide", i, guides[i], "got", a, "want", b)
                if only < 0:
                    sys.exit(1)
        # ... and as FASTQ text cut at random line ends (any phase of the four-line cycle), every part pushed on its own
        cuts = rng.randrange(0, 12)
        where = rng.choice([0, 1])
        announce = rng.random() < 0.7
        if run and reads and len(reads) <= 5000:
            import torch
            from test_ingest_gpu import _count_parts, _cut_at_lines
            text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
            parts = _cut_at_lines(text, random.Random(n_cases), cuts) or [text]
            dl = lib.device(not exact, 0, opts)
            ffi = S._ffi
            got = _count_parts(torch, S, dl, parts, reverse, o, recursion, ffi.MEM_DEVICE if where else ffi.MEM_HOST, announce=announce)
            n_checks += 1
            if got != (want, tot, mat):
                print("MISMATCH (fastq parts) seed %d case %d: L %d o %d reverse %s exact %s recursion %s opts %s parts %d where %d announce %s" % (
                    seed, n_cases, L, o, reverse, exact, recursion, opts, len(parts), where, announce))
                if only < 0:
                    sys.exit(1)
        n_cases += 1
        if only >= 0 and n_cases > only:
            break
    print("fuzz ok: seed %d, %d cases, %d comparisons in %.0f s" % (seed, n_cases, n_checks, time.time() - t0))


if __name__ == "__main__":
    main()
