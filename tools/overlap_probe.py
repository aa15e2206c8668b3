#!/usr/bin/env python3
"""Do two half-size count passes on two streams overlap on one GPU?  (feasibility probe for splitting a pass in two)
python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgcount_amd.workload import DeviceWorkload
from sgcount_amd import synth

def main():
    n = 100_000_000
    full = DeviceWorkload(n, 100_000, 20, one_mismatch=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    halves = []
    for k, st in enumerate((s1, s2)):
        with torch.cuda.stream(st):
            w = DeviceWorkload(n // 2, 100_000, 20, one_mismatch=True, reads_seed=synth.READS_SEED + 7 * k)
            w.dl.set_stream(st.cuda_stream)
            halves.append(w)
    torch.cuda.synchronize()
    def timeit(fn, reps=10):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    print("one 100M pass            : %.3f ms" % timeit(lambda: full.step()))
    print("one 50M pass             : %.3f ms" % timeit(lambda: halves[0].step()))
    def both():
        halves[0].step(); halves[1].step()
    print("two 50M passes, 2 streams: %.3f ms" % timeit(both))
    def seq():
        halves[0].step(); torch.cuda.synchronize(); halves[1].step(); torch.cuda.synchronize()
    print("two 50M passes, serial   : %.3f ms" % timeit(seq))

main()
