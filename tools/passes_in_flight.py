"""Two batches in flight: does a pass overlap with the next one when they run on two contexts / two streams?
k_partition is bound by memory, the resolver by its vector units and latency chains, and one workgroup of each fits on a CU
(LDS 79 + 58 KiB, 16 + 16 waves): alternate the passes of a resident sample between two contexts and compare with the same
number of passes on one.  usage (GPU box): python tools/passes_in_flight.py [--reads N] [--steps K] [--workload 1mm|exact]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--guides", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--workload", choices=["1mm", "exact"], default="1mm")
    ap.add_argument("--contexts", type=int, default=2)
    args = ap.parse_args()
    import torch
    from sgcount_amd.workload import DeviceWorkload
    wls = [DeviceWorkload(args.reads, args.guides, 20, one_mismatch=args.workload == "1mm")]
    streams = [torch.cuda.current_stream()]
    for k in range(1, args.contexts):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            w = DeviceWorkload(1, args.guides, 20, one_mismatch=args.workload == "1mm")
        w.n_reads, w.records = wls[0].n_reads, wls[0].records          # the same resident sample
        w.dl.set_stream(s.cuda_stream)
        wls.append(w); streams.append(s)
    torch.cuda.synchronize()

    def run(n_ctx, steps):
        for w in wls[:n_ctx]:
            w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            wls[i % n_ctx].step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    want = None
    for r in range(args.rounds):
        line = []
        for n_ctx in range(1, args.contexts + 1):
            ms = run(n_ctx, args.steps)
            line.append("%d in flight: %.3f ms per pass (%.1f G reads/s)" % (n_ctx, ms, args.reads / ms / 1e6))
        print("   ".join(line), flush=True)
    for w in wls:
        c, total, matched = w.result()
        if want is None:
            want = (c.tobytes(), total, matched)
        assert (c.tobytes(), total, matched) == want, "the contexts disagree"
    print("tables equal: total %d matched %d" % (want[1], want[2]))


if __name__ == "__main__":
    main()
