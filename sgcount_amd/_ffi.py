"""ctypes binding of libsgcount_hip.so (the C ABI in include/sgcount_hip.h).

Fails loudly: if the shared library is missing and cannot be built, or a call returns an
error code, an exception is raised — there is no CPU fallback for the count path.
"""
import ctypes as C
import os

from . import build as _build

OK, E_ARG, E_HIP, E_UNSUPPORTED, E_DUPLICATE, E_STATE, E_OOM, E_FORMAT = 0, -1, -2, -3, -4, -5, -6, -7
MEM_HOST, MEM_DEVICE = 0, 1


class SgcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sgcount_hip error {code}: {msg}")
        self.code = code


class Timing(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("lookup_ms", C.c_double), ("hist_ms", C.c_double),
                ("pack_ms", C.c_double), ("part_ms", C.c_double), ("miss_ms", C.c_double), ("h2d_ms", C.c_double)]


class LibInfo(C.Structure):
    _fields_ = [("n_guides", C.c_uint32), ("guide_len", C.c_uint32), ("record_bytes", C.c_uint32),
                ("one_mismatch", C.c_uint32), ("lib_slots", C.c_uint64), ("perm_slots", C.c_uint64),
                ("perm_entries", C.c_uint64), ("table_bytes", C.c_uint64), ("core_partitions", C.c_uint64),
                ("path", C.c_uint32), ("slices", C.c_uint32), ("slice_record_bytes", C.c_uint32), ("reserved_", C.c_uint32)]


# every symbol include/sgcount_hip.h declares: name -> (restype, argtypes)
_vp, _u8p, _u64, _u32, _i = C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
SYMBOLS = {
    "sgc_device_count": (_i, []),
    "sgc_init": (_i, [_i, C.POINTER(_vp)]),
    "sgc_free": (None, [_vp]),
    "sgc_ctx_clone": (_i, [_vp, C.POINTER(_vp)]),
    "sgc_set_stream": (_i, [_vp, _vp]),
    "sgc_get_stream": (_vp, [_vp]),
    "sgc_set_library": (_i, [_vp, _u8p, _u32, _u32, _i]),
    "sgc_library_info": (_i, [_vp, C.POINTER(LibInfo)]),
    "sgc_lookup": (_i, [_vp, _u8p, _u64, _i, _vp]),
    "sgc_record_bytes": (_u32, [_u32]),
    "sgc_pack_reads_host": (_i, [_u8p, _vp, _u64, _u32, _i, _u32, _i, _vp]),
    "sgc_pack_reads_device": (_i, [_vp, _vp, _vp, _u64, _i, _u32, _i, _vp]),
    "sgc_sample_begin": (_i, [_vp, C.POINTER(_vp), _i, _u32, _i]),
    "sgc_sample_push_packed": (_i, [_vp, _vp, _u64, _i]),
    "sgc_sample_push_packed_async": (_i, [_vp, _vp, _u64]),
    "sgc_sample_push_reads": (_i, [_vp, _u8p, _vp, _u64, _i]),
    "sgc_sample_push_windows": (_i, [_vp, _u8p, _vp, _u64, _i, C.c_uint32]),
    "sgc_sample_push_fastq": (_i, [_vp, _u8p, _u64, _i, C.POINTER(_u64)]),
    "sgc_sample_push_fastq_part": (_i, [_vp, _u8p, _u64, _i, _u64, _u64, C.POINTER(_u64)]),
    "sgc_sample_wait_uploads": (_i, [_vp, _u32]),
    "sgc_sample_sync": (_i, [_vp]),
    "sgc_sample_finish": (_i, [_vp, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "sgc_sample_flush": (_i, [_vp]),
    "sgc_sample_device_counts": (_vp, [_vp]),
    "sgc_sample_export_device": (_i, [_vp, _vp]),
    "sgc_sample_reset": (_i, [_vp]),
    "sgc_sample_free": (None, [_vp]),
    "sgc_alloc_pinned": (_vp, [C.c_size_t]),
    "sgc_free_pinned": (None, [_vp]),
    "sgc_set_option": (_i, [_vp, C.c_char_p, C.c_int64]),
    "sgc_check_host_tables": (_i, [_u8p, _u32, _u32, _i, _vp]),
    "sgc_placement_info": (_i, [_vp, _vp]),
    "sgc_timing_enable": (_i, [_vp, _i]),
    "sgc_timing_read": (_i, [_vp, C.POINTER(Timing), _i]),
    "sgc_last_error": (C.c_char_p, []),
    "sgc_version": (C.c_char_p, []),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One process may hold only ONE HIP/ROCr runtime (a second one cannot open the KFD device and
    reports "no HIP GPUs").  PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7);
    libsgcount_hip.so needs libamdhip64.so.7 too.  Loading torch's copy first (without importing torch)
    lets the dynamic loader satisfy our DT_NEEDED by SONAME and lets a later `import torch` find the
    same file, so both sides share one runtime.  Hosts without torch (C++/Rust) use /opt/rocm's."""
    if os.environ.get("SGC_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load(so=None):
    """Loads (building first if needed) libsgcount_hip.so and binds every ABI symbol.  so: another build of the same library to
    load INSTEAD (the bounds-checked libsgcount_hip_check.so of tests/test_check_gpu.py) — only before the first load of a process."""
    global _lib
    if _lib is not None:
        return _lib
    if so is not None:
        if _build.needs_build(so):
            _build.build_one(so)
    else:
        so = _build.SO
        if _build.needs_build():
            so = _build.build()
    if not os.path.exists(so):
        raise ImportError(f"{so} is missing: the HIP extension is required (no CPU fallback)")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(so)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise SgcError(rc, load().sgc_last_error().decode("utf-8", "replace"))
    return rc
