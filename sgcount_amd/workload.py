"""Device-resident synthetic workloads for bench.py and the full-size GPU tests (plumbing).

Builds the SURVEY.md §8(d) library and reads with the synthetic generator, keeps the packed records
resident in HBM as a torch tensor, and drives the C ABI on torch's current stream so that torch
events, RCCL collectives and our kernels are ordered on one stream.
"""
import ctypes as C

import numpy as np

from . import _ffi, synth
from .host import Library, Offset


def synth_library(n_guides=100_000, L=20, seed=synth.LIB_SEED):
    seqs = synth.library(n_guides, L, seed)
    order = [bytes(r) for r in seqs]
    table = {s: b"sg%06d" % i for i, s in enumerate(order)}
    return seqs, Library(table, order)


class DeviceWorkload:
    """One sample: packed records resident on the GPU + a sgc_sample to count them into."""

    def __init__(self, n_reads, n_guides=100_000, L=20, one_mismatch=True, position_recursion=True, offset=30,
                 reads_seed=synth.READS_SEED, lib_seed=synth.LIB_SEED, mode=synth.MODE_FIXED, device_index=0,
                 gen_chunk=8_000_000, reverse=False, lib_options=None):
        import torch
        self.torch = torch
        self.n_reads, self.n_guides, self.L = int(n_reads), n_guides, L
        self.offset, self.recursion, self.one_mismatch, self.reverse = offset, position_recursion, one_mismatch, reverse
        self.reads_seed, self.mode = reads_seed, mode
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        self.lib_seqs, self.library = synth_library(n_guides, L, lib_seed)
        self.dl = self.library.device(one_mismatch, device_index, lib_options)
        self.abi = self.dl.lib
        self.dl.set_stream(torch.cuda.current_stream().cuda_stream)
        self.words = self.dl.record_bytes // 8
        self.lib_dev = torch.from_numpy(self.lib_seqs).to(self.dev)
        self.records = torch.empty(self.n_reads * self.words, dtype=torch.int64, device=self.dev)
        done = 0
        while done < self.n_reads:
            m = min(gen_chunk, self.n_reads - done)
            raw, offs = synth.reads_device(self.lib_dev, done, m, reads_seed, mode)
            _ffi.check(self.abi.sgc_pack_reads_device(self.dl.ctx, raw.data_ptr(), offs.data_ptr(), m, int(reverse),
                                                      offset, int(position_recursion),
                                                      self.records.data_ptr() + done * self.words * 8))
            torch.cuda.synchronize()
            del raw, offs
            done += m
        self.sample = C.c_void_p()
        _ffi.check(self.abi.sgc_sample_begin(self.dl.ctx, C.byref(self.sample), int(reverse), offset,
                                             int(position_recursion)))
        self.export = torch.zeros(n_guides + 2, dtype=torch.int64, device=self.dev)

    # one pass of the hot path over the whole resident sample; leaves counts|total|matched in self.export
    def step(self, first=0, n=None, out=None):
        """out: an int64 device tensor of n_guides + 2 words to receive the row instead of self.export"""
        n = self.n_reads - first if n is None else n
        _ffi.check(self.abi.sgc_sample_reset(self.sample))
        _ffi.check(self.abi.sgc_sample_push_packed(self.sample, self.records.data_ptr() + first * self.words * 8, n,
                                                   _ffi.MEM_DEVICE))
        _ffi.check(self.abi.sgc_sample_export_device(self.sample, (self.export if out is None else out).data_ptr()))

    def result(self, row=None):
        """(counts np.uint64[n_guides], total, matched) of the last step (synchronises)."""
        self.torch.cuda.synchronize()
        host = (self.export if row is None else row).cpu().numpy().view(np.uint64)
        return host[: self.n_guides].copy(), int(host[self.n_guides]), int(host[self.n_guides + 1])

    def close(self):
        if self.sample:
            self.abi.sgc_sample_free(self.sample)
            self.sample = None


def oracle_offset(offset, reverse=False):
    return Offset.Reverse(offset) if reverse else Offset.Forward(offset)
