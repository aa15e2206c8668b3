#!/usr/bin/env python3
"""End-to-end timing of the sgcount-hip command line on a synthetic FASTQ file (page cache -> count table).
python tools/e2e.py --reads 5000000   (the CPU baseline is timed by bench.py, the only place besides tests/ that may load the oracle)"""
import argparse
import gzip
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--guides", type=int, default=100_000)
    ap.add_argument("--dir", default="/tmp/sgc_e2e")
    args = ap.parse_args()
    from sgcount_amd import hostlib, synth
    os.makedirs(args.dir, exist_ok=True)
    lib = synth.library(args.guides, 20)
    lib_path = os.path.join(args.dir, "library.fa")
    open(lib_path, "wb").write(synth.library_fasta(lib))
    fq = os.path.join(args.dir, "reads.fastq")
    t0 = time.time()
    with open(fq, "wb") as f:
        for first in range(0, args.reads, 500_000):
            f.write(synth.fastq_host(lib, first, min(500_000, args.reads - first)))
    size = os.path.getsize(fq)
    print("wrote %s: %.2f GB in %.1f s" % (fq, size / 1e9, time.time() - t0), flush=True)
    gz = fq + ".gz"
    t0 = time.time()
    subprocess.check_call("gzip -1 -c %s > %s" % (fq, gz), shell=True)
    print("gzip -1: %.2f GB in %.1f s" % (os.path.getsize(gz) / 1e9, time.time() - t0), flush=True)
    cli = hostlib.cli_path()
    outs = {}
    for label, path, extra in (("plain, GPU-parsed FASTQ", fq, ["--pack", "fastq"]),
                               ("plain, host-parsed + GPU pack", fq, ["--pack", "device"]),
                               ("plain, host-parsed + host pack", fq, ["--pack", "host"]),
                               (".gz,   GPU-parsed FASTQ", gz, ["--pack", "fastq"]),
                               ("plain, GPU-parsed, exact (-x)", fq, ["--pack", "fastq", "-x"])):
        best = None
        for rep in range(2):
            out = os.path.join(args.dir, "out.tsv")
            t0 = time.time()
            subprocess.check_call([cli, "-l", lib_path, "-i", path, "-a", "30", "-q", "-o", out] + extra)
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        outs[label] = open(out).read()
        print("%-34s %.2f s  -> %.1f M reads/s end to end (process start to table, incl. table build)" % (
            label, best, args.reads / best / 1e6), flush=True)
    # several .gz samples on one GPU: -t N gives every worker its own context, so inflate/parse of one sample overlaps
    # the others' (the reference's rayon pool over samples)
    k = 8
    links = []
    for i in range(k):
        ln = os.path.join(args.dir, "s%d.fastq.gz" % i)
        if os.path.lexists(ln):
            os.remove(ln)
        os.symlink(gz, ln)
        links.append(ln)
    for t in (1, k):
        out = os.path.join(args.dir, "out_multi.tsv")
        t0 = time.time()
        subprocess.check_call([cli, "-l", lib_path, "-i"] + links + ["-a", "30", "-q", "-o", out, "-t", str(t)])
        dt = time.time() - t0
        print("%d x .gz samples, -t %-2d               %.2f s  -> %.1f M reads/s end to end" % (k, t, dt, k * args.reads / dt / 1e6),
              flush=True)
    assert outs["plain, GPU-parsed FASTQ"] == outs["plain, host-parsed + GPU pack"] == outs["plain, host-parsed + host pack"] \
        == outs[".gz,   GPU-parsed FASTQ"]


if __name__ == "__main__":
    main()
