#!/usr/bin/env python3
"""Prints the last N count-path kernel launches of the newest rocprofv3 kernel trace under gpurun_out/<name>/ (duration, gap)."""
import csv, glob, os, sys
name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = max(glob.glob('gpurun_out/%s/*/*kernel_trace.csv' % name), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_cp', 'k_core', 'k_partition', 'k_count_slices', 'k_export', 'k_resolve', 'k_hist', 'k_generic'))]
prev = None
for r in sel[-n:]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-44s %8.1f us  gap %6.1f" % (r['Kernel_Name'][:44], (en - st) / 1e3, (st - prev) / 1e3 if prev else 0))
    prev = en
