#!/usr/bin/env python3
"""ONE process, several pool allocations: K1/K2/A/B of a 100M-read pass for each way of allocating the block pool.
python tools/regime_sweep.py "vmm_chunk_mb=0" "vmm_chunk_mb=32" "vmm_chunk_mb=2,vmm_shuffle=3" ...  (each spec applied, scratch dropped, 3 warm + 10 timed passes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402,F401
from sgcount_amd.workload import DeviceWorkload         # noqa: E402

wl = DeviceWorkload(100_000_000, 100_000, 20, one_mismatch=True)
for spec in sys.argv[1:]:
    keep = False
    for kv in filter(None, spec.split(",")):
        if kv == "keep":                      # same scratch allocations as the spec before: the regime stays, only the options change
            keep = True
            continue
        k, v = kv.split("=")
        wl.dl.set_option(k, int(v))
    if not keep:
        wl.dl.set_option("drop_scratch", 1)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    wl.dl.timing(True)
    wl.dl.timing(reset=True)
    steps = 10
    for _ in range(steps):
        wl.step()
    t = wl.dl.timing(reset=True)
    wl.dl.timing(False)
    print("%-40s K1 %.3f  K2 %.3f  A %.3f  B %.3f  = %.3f ms" % (spec, t.part_ms / steps, t.lookup_ms / steps, t.miss_ms / steps, t.hist_ms / steps,
                                                                (t.part_ms + t.lookup_ms + t.miss_ms + t.hist_ms) / steps), flush=True)
wl.close()
