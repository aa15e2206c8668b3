#!/usr/bin/env python3
"""Times the on-device ingest paths on synthetic data resident in HBM:
   raw reads (bytes + offsets) -> records   (sgc_pack_reads_device)
   FASTQ text                  -> records   (sgc_sample_push_fastq_part: k_fastq_count + k_scan_tiles + k_fastq_pack)
and the whole FASTQ -> counts path.  python tools/tune_ingest.py --reads 20000000"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--guides", type=int, default=100_000)
    ap.add_argument("--chunk", type=int, default=10_000_000)
    ap.add_argument("--dbg", type=int, default=0, help="timing-only ablation flags of k_fastq_pack (<< 24 into the ctx dbg word)")
    args = ap.parse_args()
    import torch
    from sgcount_amd import _ffi, synth
    from sgcount_amd.workload import synth_library
    lib_seqs, library = synth_library(args.guides, 20)
    dl = library.device(True)
    dl.set_stream(torch.cuda.current_stream().cuda_stream)
    lib_dev = torch.from_numpy(lib_seqs).cuda()
    smp = C.c_void_p()
    _ffi.check(dl.lib.sgc_sample_begin(dl.ctx, C.byref(smp), 0, 30, 1))
    dl.timing(True)
    if args.dbg:
        dl.set_option("dbg", args.dbg << 24)
    for what in ("reads", "fastq"):
        tot_bytes, tot_ms, wall_ms = 0, 0.0, 0.0
        for first in range(0, args.reads, args.chunk):
            m = min(args.chunk, args.reads - first)
            if what == "reads":
                raw, offs = synth.reads_device(lib_dev, first, m)
                recs = torch.empty(m, dtype=torch.int64, device="cuda")
                for rep in range(3):
                    dl.timing(reset=True)
                    _ffi.check(dl.lib.sgc_pack_reads_device(dl.ctx, raw.data_ptr(), offs.data_ptr(), m, 0, 30, 1, recs.data_ptr()))
                    t = dl.timing(reset=True)
                tot_bytes += raw.numel(); tot_ms += t.pack_ms
            else:
                fq, _ = synth.fastq_device(lib_dev, first, m)
                n_nl = int((fq == 10).sum().item())
                for rep in range(3):
                    _ffi.check(dl.lib.sgc_sample_reset(smp))
                    dl.timing(reset=True)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    n = C.c_uint64(0)
                    _ffi.check(dl.lib.sgc_sample_push_fastq_part(smp, fq.data_ptr(), fq.numel(), _ffi.MEM_DEVICE, 0, n_nl, C.byref(n)))
                    e1.record()
                    torch.cuda.synchronize()
                    t = dl.timing(reset=True)
                assert n.value == m or args.dbg
                tot_bytes += fq.numel(); tot_ms += t.pack_ms; wall_ms += e0.elapsed_time(e1)
                del fq
        print("%-6s -> records: %.2f GB in %.2f ms = %.2f TB/s, %.1f G reads/s" % (
            what, tot_bytes / 1e9, tot_ms, tot_bytes / tot_ms / 1e9, args.reads / tot_ms / 1e6) +
              ("" if what == "reads" else "   | text -> counts %.2f ms = %.1f G reads/s" % (wall_ms, args.reads / wall_ms / 1e6)))
    dl.lib.sgc_sample_free(smp)


if __name__ == "__main__":
    main()
