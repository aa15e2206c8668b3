#!/usr/bin/env python3
"""One process = one sample of the pass regimes: K1/K2/core times of a 100M-read pass, with ctx options k=v ... applied before the
first pass.   for i in 1 2 3 4; do python tools/regime_runs.py contig_pool=1; done"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402,F401
from sgcount_amd.workload import DeviceWorkload         # noqa: E402

opts = dict(kv.split("=") for kv in sys.argv[1:])
reads = int(opts.pop("reads", 100_000_000))
wl = DeviceWorkload(reads, 100_000, 20, one_mismatch=True)
for k, v in opts.items():
    wl.dl.set_option(k, int(v))
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
wl.dl.timing(True)
wl.dl.timing(reset=True)
steps = 10
for _ in range(steps):
    wl.step()
t = wl.dl.timing(reset=True)
print("%-28s K1 %.3f  K2 %.3f  A %.3f  B %.3f  = %.3f ms" % (" ".join(sys.argv[1:]) or "(default)", t.part_ms / steps, t.lookup_ms / steps,
                                                             t.miss_ms / steps, t.hist_ms / steps, (t.part_ms + t.lookup_ms + t.miss_ms + t.hist_ms) / steps), flush=True)
wl.close()
