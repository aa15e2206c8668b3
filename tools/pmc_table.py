#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: one row per kernel, mean per dispatch."""
import collections
import csv
import sys

want = sys.argv[2:] if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    if want and not any(w in name for w in want):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-24s n=%-3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
