"""ctypes binding of libsgcount_synth.so — the synthetic workload of SURVEY.md §8(d).

Bench/test plumbing, not part of the reference's interface.  Host generators return numpy
arrays / bytes; device generators fill torch CUDA tensors (byte-identical to the host ones).
"""
import ctypes as C

import numpy as np

from . import _ffi, build as _build

MODE_FIXED, MODE_STAGGER = 0, 1


def mode_dominant(pct):
    """| into a mode: pct percent of the reads draw ONE guide (include/sgcount_synth.h SGS_MODE_DOMINANT)"""
    return (int(pct) & 127) << 8
READ_LEN, PREFIX_LEN = 150, 30
LIB_SEED, READS_SEED = 0x5EED0001, 0x5EED0002

_vp, _u64, _u32, _sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
SYMBOLS = {
    "sgs_library": (C.c_int, [_u64, _u32, _u32, _vp]),
    "sgs_library_fasta": (_sz, [_vp, _u32, _u32, _vp, _sz]),
    "sgs_read_len": (_u32, [_u64, _u64, _u32, _u32]),
    "sgs_read_class": (_u32, [_u64, _u64, _u32, _u32, _u32, C.POINTER(_u32)]),
    "sgs_reads_host": (C.c_int, [_u64, _u64, _u64, _vp, _u32, _u32, _u32, _vp, _vp]),
    "sgs_fastq_host": (_sz, [_u64, _u64, _u64, _vp, _u32, _u32, _u32, _vp, _sz]),
    "sgs_read_lens_device": (C.c_int, [_vp, _u64, _u64, _u64, _u32, _u32, _vp]),
    "sgs_reads_device": (C.c_int, [_vp, _u64, _u64, _u64, _vp, _u32, _u32, _u32, _vp, _vp]),
    "sgs_fastq_lens_device": (C.c_int, [_vp, _u64, _u64, _u64, _u32, _u32, _vp]),
    "sgs_fastq_device": (C.c_int, [_vp, _u64, _u64, _u64, _vp, _u32, _u32, _u32, _vp, _vp]),
    "sgs_last_error": (C.c_char_p, []),
}
_lib = None


def load():
    global _lib
    if _lib is None:
        if _build.needs_build(_build.SYNTH_SO):
            _build.build_one(_build.SYNTH_SO)
        _ffi._share_hip_runtime_with_torch()
        lib = C.CDLL(_build.SYNTH_SO)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("sgcount_synth: " + load().sgs_last_error().decode())


def library(n=100_000, L=20, seed=LIB_SEED):
    """→ uint8 array (n, L) of ASCII guides."""
    out = np.empty((n, L), dtype=np.uint8)
    _chk(load().sgs_library(seed, n, L, out.ctypes.data))
    return out


def library_fasta(seqs: np.ndarray) -> bytes:
    n, L = seqs.shape
    seqs = np.ascontiguousarray(seqs)
    need = load().sgs_library_fasta(seqs.ctypes.data, n, L, None, 0)
    buf = np.empty(need, dtype=np.uint8)
    got = load().sgs_library_fasta(seqs.ctypes.data, n, L, buf.ctypes.data, need)
    return buf[:got].tobytes()


def reads_host(lib_seqs: np.ndarray, first, n, seed=READS_SEED, mode=MODE_FIXED):
    """→ (uint8 bytes, uint64 offsets[n+1]) for reads [first, first+n)."""
    ng, L = lib_seqs.shape
    lib_seqs = np.ascontiguousarray(lib_seqs)
    seqs = np.empty(n * READ_LEN, dtype=np.uint8)
    offs = np.empty(n + 1, dtype=np.uint64)
    _chk(load().sgs_reads_host(seed, first, n, lib_seqs.ctypes.data, ng, L, mode, seqs.ctypes.data, offs.ctypes.data))
    return seqs[: int(offs[n])], offs


def fastq_host(lib_seqs: np.ndarray, first, n, seed=READS_SEED, mode=MODE_FIXED) -> bytes:
    ng, L = lib_seqs.shape
    lib_seqs = np.ascontiguousarray(lib_seqs)
    need = load().sgs_fastq_host(seed, first, n, lib_seqs.ctypes.data, ng, L, mode, None, 0)
    buf = np.empty(need, dtype=np.uint8)
    got = load().sgs_fastq_host(seed, first, n, lib_seqs.ctypes.data, ng, L, mode, buf.ctypes.data, need)
    assert got == need
    return buf.tobytes()


def read_class(i, n_guides, L=20, seed=READS_SEED, mode=MODE_FIXED):
    g = C.c_uint32(0)
    c = load().sgs_read_class(seed, i, n_guides, L, mode, C.byref(g))
    return c, g.value


def _stream_ptr(torch):
    return torch.cuda.current_stream().cuda_stream


def reads_device(lib_seqs_dev, first, n, seed=READS_SEED, mode=MODE_FIXED):
    """lib_seqs_dev: torch uint8 CUDA tensor (n_guides, L).  → (uint8 CUDA bytes, int64 CUDA offsets[n+1])."""
    import torch
    ng, L = lib_seqs_dev.shape
    st = _stream_ptr(torch)
    lens = torch.empty(n, dtype=torch.int32, device=lib_seqs_dev.device)
    _chk(load().sgs_read_lens_device(st, seed, first, n, L, mode, lens.data_ptr()))
    offs = torch.zeros(n + 1, dtype=torch.int64, device=lib_seqs_dev.device)
    torch.cumsum(lens, 0, out=offs[1:])
    total = int(offs[-1].item())
    out = torch.empty(total, dtype=torch.uint8, device=lib_seqs_dev.device)
    _chk(load().sgs_reads_device(st, seed, first, n, lib_seqs_dev.data_ptr(), ng, L, mode, offs.data_ptr(),
                                 out.data_ptr()))
    return out, offs


def fastq_device(lib_seqs_dev, first, n, seed=READS_SEED, mode=MODE_FIXED):
    """→ (uint8 CUDA FASTQ text, int64 CUDA record offsets[n+1])."""
    import torch
    ng, L = lib_seqs_dev.shape
    st = _stream_ptr(torch)
    lens = torch.empty(n, dtype=torch.int32, device=lib_seqs_dev.device)
    _chk(load().sgs_fastq_lens_device(st, seed, first, n, L, mode, lens.data_ptr()))
    offs = torch.zeros(n + 1, dtype=torch.int64, device=lib_seqs_dev.device)
    torch.cumsum(lens, 0, out=offs[1:])
    total = int(offs[-1].item())
    out = torch.empty(total, dtype=torch.uint8, device=lib_seqs_dev.device)
    _chk(load().sgs_fastq_device(st, seed, first, n, lib_seqs_dev.data_ptr(), ng, L, mode, offs.data_ptr(),
                                 out.data_ptr()))
    return out, offs
