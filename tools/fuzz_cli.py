#!/usr/bin/env python3
"""Randomised differential run of the command line on the GPU box: random libraries (some with guides outside ACGT), random FASTQ
samples written plain / gzip (one or several members) / BGZF, with LF or CRLF, with or without a final newline, with trailing blank
lines; random --pack, scanner threads and block sizes, -x, -p, -r; the table against the CPU oracle's.

    python3 tools/fuzz_cli.py [seconds] [seed]"""
import gzip
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F                      # noqa: E402
from test_cli_gpu import oracle_table         # noqa: E402
import _oracle as O                          # noqa: E402
from sgcount_amd import hostlib               # noqa: E402
from sgcount_amd.bgzf import bgzf_bytes       # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    cli = hostlib.cli_path()
    d = tempfile.mkdtemp()
    t0, n_cases = time.time(), 0
    last_note = time.time()
    while time.time() - t0 < seconds:
        if time.time() - last_note > 60:
            print("... %d cases so far" % n_cases, flush=True); last_note = time.time()
        L, o, guides, reads = F.case(rng)
        reads = [r for r in reads if b"\n" not in r]
        if not reads or max(len(r) for r in reads) < L:       # count.rs:98-100 refuses a sample whose first read is shorter than the guides
            continue
        if len(reads[0]) < L:
            reads[0] = reads[0] + b"A" * (L - len(reads[0]))
        # (an empty LAST read is drawn like any other: a stream that ends behind a separator line ends with a record whose quality line
        # is empty — reader decision #3; round 3 kept this case out of the draw)
        reverse = rng.random() < 0.4
        if reverse:
            reads = [bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(r)) for r in reads]
        exact, recursion = rng.random() < 0.3, rng.random() < 0.7
        eol = b"\r\n" if rng.random() < 0.2 else b"\n"
        text = b"".join(b"@r%d%s%s%s+%s%s%s" % (i, eol, r, eol, eol, bytes(rng.choice(b"@+I#5") for _ in range(len(r))), eol) for i, r in enumerate(reads))
        plain = text
        u = rng.random()
        if u < 0.15:
            text = text[: -len(eol)]                        # no final newline
        elif u < 0.3:
            text += b"\n" * rng.choice([1, 2, 5])           # trailing blank lines
        kind = rng.choice(["plain", "gz", "gzm", "bgzf"])
        path = os.path.join(d, "s.fastq" + ("" if kind == "plain" else ".gz"))
        if kind == "plain":
            blob = text
        elif kind == "gz":
            blob = gzip.compress(text, rng.choice([1, 6, 9]))
        elif kind == "gzm":
            cuts = sorted(rng.randrange(len(text) + 1) for _ in range(rng.choice([1, 2, 4])))
            blob = b"".join(gzip.compress(text[a:b], rng.choice([1, 6])) for a, b in zip([0] + cuts, cuts + [len(text)]))
        else:
            blob = bgzf_bytes(text, rng.choice([3000, 20000, 65280]), eof_marker=rng.random() < 0.7)
        open(path, "wb").write(blob)
        lib_text = b"".join(b">g%d\n%s\n" % (i, g) for i, g in enumerate(guides))
        lp = os.path.join(d, "lib.fa")
        open(lp, "wb").write(lib_text)
        oracle_text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
        use = (reverse, o)
        auto = rng.random() < 0.25 and not reverse
        if auto:
            # no -a: the entropy offsetter picks the offset (src/offsetter.rs); the oracle's choice is the expectation
            try:
                use = O.entropy_offset(lib_text, oracle_text)
            except O.OracleError:
                auto = False
        args = [cli, "-l", lp, "-i", path] + ([] if auto else ["-a", str(o), "-q"])
        if exact: args.append("-x")
        if not recursion: args.append("-p")
        if reverse: args.append("-r")
        args += ["--pack", rng.choice(["scan", "scan", "scan", "fastq", "device", "host"])]
        args += ["--scan-threads", str(rng.choice([1, 2, 3, 7]))]
        args += ["--scan-block-kb", str(rng.choice([1, 4, 64, 4096]))]
        p = subprocess.run(args, capture_output=True, timeout=300)
        # the oracle reads what the reference would: CRLF lines end with '\r' for fxread too?  It strips them (as the scanner does), so
        # the oracle gets the reads themselves
        want = oracle_table(lib_text, [oracle_text], ["s"], [use], exact, recursion)
        said = "Calculated Offsets: [%s(%d)]" % ("Reverse" if use[0] else "Forward", use[1])
        if p.returncode != 0 or p.stdout.decode() != want or (auto and said not in p.stderr.decode()):
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_cli_fail")
            os.makedirs(keep, exist_ok=True)
            open(os.path.join(keep, os.path.basename(path)), "wb").write(blob)
            open(os.path.join(keep, "lib.fa"), "wb").write(lib_text)
            open(os.path.join(keep, "want.tsv"), "w").write(want)
            open(os.path.join(keep, "got.tsv"), "wb").write(p.stdout)
            print("MISMATCH seed %d case %d: rc %d kind %s eol %r auto %s expected offset %s args %s\nstderr: %s" % (seed, n_cases, p.returncode, kind, eol, auto, use, " ".join(args[1:]), p.stderr.decode()[-900:]))
            sys.exit(1)
        n_cases += 1
    print("cli fuzz ok: seed %d, %d cases in %.0f s" % (seed, n_cases, time.time() - t0))


if __name__ == "__main__":
    main()
