#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/pmc_traffic.json.

    python tools/pmc_traffic.py <fetch_dir_1mm> <write_dir_1mm> <fetch_dir_exact> <write_dir_exact> [round-label]

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reads exactly half the bytes
of a wide coalesced streaming read (128-B requests tallied at 64 B); other access widths are uncalibrated, so
the factor is calibrated here on a kernel of the same run whose fetched byte count is known exactly:
k_partition reads every 8-byte record once (n_reads x 8 B) and nothing else of size.
Per-pass traffic = Σ over the count-pipeline kernels of one step.
"""
import collections
import csv
import glob
import json
import os
import sys

PIPE = ("k_partition", "k_count_slices", "k_cp_count", "k_cp_scatter", "k_core", "k_generic", "k_resolve_miss", "k_hist_segments",
        "k_export")


def per_kernel(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[name].append(float(r["Counter_Value"]))
    steps = len(acc["k_export"])                                  # one k_export launch per step
    # k_partition / k_count_slices also run in the placement trials of the first pass (before any k_core): only their last
    # `steps` dispatches belong to steps
    for k in ("k_partition", "k_count_slices"):
        if k in acc:
            acc[k] = acc[k][-steps:]
    return {k: sum(v) / steps for k, v in acc.items()}            # per step (a kernel may run twice in a step)


def main():
    out = {}
    n_reads = 100_000_000
    for w, fd, wd in (("1mm", sys.argv[1], sys.argv[2]), ("exact", sys.argv[3], sys.argv[4])):
        fetch, write = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
        cal = (n_reads * 8) / (fetch["k_partition"] * 1024.0)      # true bytes / reported bytes for our 8-B/lane stream
        kern = {}
        tot = 0.0
        for k in PIPE:
            if k not in fetch:
                continue
            rb, wb = fetch[k] * 1024.0 * cal, write.get(k, 0.0) * 1024.0
            kern[k] = {"fetch_bytes_raw": fetch[k] * 1024.0, "fetch_bytes_corrected": rb, "write_bytes": wb}
            tot += rb + wb
        out[w] = {"hbm_bytes_per_step": tot, "bytes_per_read": tot / n_reads, "fetch_correction": cal,
                  "collected": "profiles/%s" % (sys.argv[5] if len(sys.argv) > 5 else "?"),
                  "unit": "bytes per 100M-read pass (FETCH_SIZE x correction + WRITE_SIZE)", "kernels": kern}
    # which kernels these bytes belong to: bench.py quotes roofline.traffic only while the device sources are these
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_sha16
    out["kernel_sources_sha16"] = kernel_sources_sha16()
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                     "pmc_traffic.json"), "w"), indent=1)
    for w in ("1mm", "exact"):
        print(w, "%.2f GB per pass, %.1f B/read, fetch correction x%.2f" % (out[w]["hbm_bytes_per_step"] / 1e9,
                                                                             out[w]["bytes_per_read"], out[w]["fetch_correction"]))
        for k, v in out[w]["kernels"].items():
            print("   %-18s fetch %.3f GB (raw %.3f)  write %.3f GB" % (k, v["fetch_bytes_corrected"] / 1e9,
                                                                          v["fetch_bytes_raw"] / 1e9, v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main()
