#!/usr/bin/env python3
"""Per-kernel averages over the TIMED region of a bench.py run under rocprofv3 --kernel-trace: the last <steps> dispatches of every
count-pipeline kernel (the warm-up steps and, with them, the K1 + K2 dispatches of the placement trials come first and are left out —
rocprofv3's own --stats averages over all dispatches).  usage: tools/trace_timed.py <kernel_trace.csv> <steps>   -> csv on stdout"""
import collections
import csv
import sys

steps = int(sys.argv[2])
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if any(k in name for k in ("k_partition", "k_count_slices", "k_core", "k_export")):
        by[name.split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
w = csv.writer(sys.stdout)
w.writerow(["Name", "CallsInTimedRegion", "AverageNs", "MinNs", "MaxNs", "AllCalls"])
tot = 0.0
for name, d in sorted(by.items()):
    t = d[-steps:]
    tot += sum(t) / len(t)
    w.writerow([name, len(t), "%.1f" % (sum(t) / len(t)), min(t), max(t), len(d)])
w.writerow(["SUM of the averages (one pass)", "", "%.1f" % tot, "", "", ""])
