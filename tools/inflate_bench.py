#!/usr/bin/env python3
"""Throughput of the gzip inflaters of the text path on a FASTQ-like file: zlib on one thread against the parallel decoder.
python tools/inflate_bench.py [reads] [threads ...]   (no GPU)"""
import gzip
import os
import random
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcount_amd import hostlib          # noqa: E402


def main():
    synth_text = "--synth" in sys.argv          # the bench's synthetic FASTQ (constant adapters, qualities all 'I': ratio ~18)
    argv = [a for a in sys.argv[1:] if a != "--synth"]
    reads = int(argv[0]) if argv else 2_000_000
    threads = [int(x) for x in argv[1:]] or [1, 2, 4, 8]
    d = "/dev/shm/sgc_inflate_%d" % os.getpid()
    os.makedirs(d, exist_ok=True)
    try:
        r = random.Random(5)
        fq = os.path.join(d, "r.fastq")
        if synth_text:
            from sgcount_amd import synth
            lib = synth.library(100_000, 20)
            with open(fq, "wb") as f:
                for first in range(0, reads, 500_000):
                    f.write(synth.fastq_host(lib, first, min(500_000, reads - first)))
        else:
          with open(fq, "wb") as f:
            bases = [bytes(r.choice(b"ACGT") for _ in range(150)) for _ in range(4096)]
            quals = [bytes(r.choice(b"FFFFFFFF:,#") for _ in range(150)) for _ in range(4096)]
            for i in range(reads):
                # shuffled pieces of random reads: compresses about as well as real data (ratio ~3.5-4.5)
                s = bases[r.randrange(4096)][: 75] + bases[r.randrange(4096)][75:]
                f.write(b"@SRR1.%d %d/1\n%s\n+\n%s\n" % (i, i, s, quals[r.randrange(4096)]))
        size = os.path.getsize(fq)
        for level in (1, 6):
            gz = os.path.join(d, "r%d.fastq.gz" % level)
            subprocess.check_call("gzip -%d -c %s > %s" % (level, fq, gz), shell=True)
            print("gzip -%d: %.1f MB -> %.1f MB (ratio %.2f)" % (level, size / 1e6, os.path.getsize(gz) / 1e6, size / os.path.getsize(gz)), flush=True)
            for t in threads:
                t0 = time.perf_counter()
                n, lines, busy, fb, kind = hostlib.text_feeder_drain(gz, 32 << 20, t)
                dt = time.perf_counter() - t0
                assert n == size and lines == 4 * reads
                print("   %-4s %2d threads: %.2f s  %.0f MB/s of text  (busy %.2f s, %d in-order chunks)" % (kind, t, dt, size / dt / 1e6, busy, fb), flush=True)
    finally:
        subprocess.call(["rm", "-rf", d])


main()
