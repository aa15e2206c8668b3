#!/usr/bin/env python3
"""A/B timing of count-kernel variants in ONE process (interleaved rounds), on the bench workload.

    python tools/tune.py --reads 100000000 --variants 4,3 --rounds 3 --workload 1mm
Prints per variant: median/min of Σ kernel time per pass (lookup, hist), and checks every variant
returns the same count table.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# dbg=... ablation flags need a library built with SGC_HIPCC_FLAGS=-DSGC_ABLATE=1 (python -c "from sgcount_amd import build as b; b.build_one(b.SO, force=True)")


def synth_mode(dominant):
    from sgcount_amd import synth
    return synth.MODE_FIXED | synth.mode_dominant(dominant)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--guides", type=int, default=100_000)
    ap.add_argument("--variants", default="4")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--workload", choices=["1mm", "exact"], default="1mm")
    ap.add_argument("--nocheck", action="store_true")
    ap.add_argument("--dominant", type=int, default=0, help="percent of the reads that draw ONE guide (synth.mode_dominant)")
    ap.add_argument("--notiming", action="store_true", help="leave the library's per-kernel events off (wall time only)")
    ap.add_argument("--opts", default="", help="extra options k=v,k=v applied to all variants")
    ap.add_argument("--lib-opts", default="", help="options k=v,k=v applied BEFORE the tables are built (align_slices, rest_filter, ...)")
    ap.add_argument("--lib", choices=["ablate", "stamps"], default=None, help="load an experiment build of the library (sgcount_amd/build.py EXPERIMENT_SOS; built on the spot if stale)")
    ap.add_argument("--lib-path", default=None, help="load this build of the library instead (an experiment .so under sgcount_amd/)")
    args = ap.parse_args()
    if args.lib_path:
        from sgcount_amd import _ffi, build as _b
        _b.TARGETS.setdefault(os.path.abspath(args.lib_path), _b.TARGETS[_b.SO])
        _ffi.load(os.path.abspath(args.lib_path))
    if args.lib:
        from sgcount_amd import _ffi, build as _b
        _ffi.load(_b.EXPERIMENT_SOS[args.lib])
    import torch
    from sgcount_amd.workload import DeviceWorkload
    lib_opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in filter(None, args.lib_opts.split(","))}
    wl = DeviceWorkload(args.reads, args.guides, 20, one_mismatch=args.workload == "1mm", lib_options=lib_opts,
                        **({"mode": synth_mode(args.dominant)} if args.dominant else {}))
    for kv in filter(None, args.opts.split(",")):
        k, v = kv.split("=")
        wl.dl.set_option(k, int(v))
    specs = {}
    variants = []
    for item in args.variants.split(","):
        v, _, opts = item.partition(":")
        variants.append(item)
        specs[item] = (int(v), [kv.split("=") for kv in filter(None, opts.split(";"))])
    res = {v: {"lookup": [], "hist": [], "wall": [], "part": [], "miss": []} for v in variants}
    sig = {}
    wl.dl.timing(not args.notiming)
    for r in range(args.rounds + 1):
        for v in variants:
            wl.dl.set_option("variant", specs[v][0])
            wl.dl.set_option("dbg", 0)
            for k in ("cuckoo", "dense", "tag_sub", "direct", "six_byte", "five_byte", "balanced", "wide"):          # (place_trials is per context: --opts place_trials=1 turns the trials off)          # per-variant toggles start from their defaults
                wl.dl.set_option(k, 1)
            for k, val in specs[v][1]:
                wl.dl.set_option(k, int(val))
            torch.cuda.synchronize()
            wl.dl.timing(reset=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                wl.step()
            e1.record()
            torch.cuda.synchronize()
            if any(k == "dbg" and int(val) & 1048576 for k, val in specs[v][1]):
                sys.stdout.flush()
                wl.dl.set_option("timeline_dump", 1)       # a -DSGC_STAMPS=1 library: "TL" lines of the last pass (tools/wg_timeline.py)
            t = wl.dl.timing(reset=True)
            counts, total, matched = wl.result()
            h = hashlib.sha256(counts.tobytes()).hexdigest()[:12] + ":%d:%d" % (total, matched)
            sig.setdefault(v, h)
            assert args.nocheck or sig[v] == h
            if r == 0:
                continue  # warm-up round
            res[v]["lookup"].append(t.lookup_ms / args.steps)
            res[v]["hist"].append(t.hist_ms / args.steps)
            res[v]["part"].append(t.part_ms / args.steps)
            res[v]["miss"].append(t.miss_ms / args.steps)
            res[v]["wall"].append(e0.elapsed_time(e1) / args.steps)
    wl.dl.set_option("print_occupancy", 1)         # where the scratch buffers fell (stderr)
    assert args.nocheck or len(set(sig.values())) == 1, "variants disagree: %r" % sig
    print("reads=%d workload=%s table=%s" % (args.reads, args.workload, next(iter(sig.values()))))
    for v in variants:
        d = res[v]
        print("variant %-22s part %.3f  lookup %.3f (min %.3f)  miss %.3f  hist %.3f (min %.3f)  wall/step %.3f (min %.3f) ms  -> %.1f Greads/s" % (
            v, statistics.median(d["part"]), statistics.median(d["lookup"]), min(d["lookup"]), statistics.median(d["miss"]),
            statistics.median(d["hist"]), min(d["hist"]),
            statistics.median(d["wall"]), min(d["wall"]), args.reads / statistics.median(d["wall"]) / 1e6))


if __name__ == "__main__":
    main()
