#!/usr/bin/env python3
"""Times sgc_init + sgc_set_library (100k guides) cold and warm, 1mm and exact: where the CLI's table_build_s goes."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcount_amd import _ffi                       # noqa: E402
from sgcount_amd.workload import synth_library     # noqa: E402

t0 = time.perf_counter()
lib = _ffi.load()
seqs, library = synth_library(100_000, 20)
flat = seqs.tobytes()
print("load + synth %.3f s" % (time.perf_counter() - t0))
for rep in range(3):
    for one_mm in (1, 0):
        ctx = C.c_void_p()
        t0 = time.perf_counter()
        _ffi.check(lib.sgc_init(0, C.byref(ctx)))
        if rep == 2:
            _ffi.check(lib.sgc_set_option(ctx, b"verbose", 1))
        t1 = time.perf_counter()
        _ffi.check(lib.sgc_set_library(ctx, flat, 100_000, 20, one_mm))
        t2 = time.perf_counter()
        print("rep %d one_mm %d: init %.3f s, set_library %.3f s" % (rep, one_mm, t1 - t0, t2 - t1))
        lib.sgc_free(ctx)
