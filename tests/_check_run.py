"""Run by tests/test_check_gpu.py in a process of its own: loads the bounds-checked build of the library (-DSGC_CHECK=1:
libsgcount_hip_check.so) INSTEAD of the shipped one and takes the partitioned pass through its rungs and sizes.  Any scratch
index out of bounds surfaces as SGC_E_STATE from sgc_sample_finish (the access is skipped, nothing faults)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgcount_amd import _ffi, build            # noqa: E402

lib = _ffi.load(build.CHECK_SO)
assert b"bounds-checked" in lib.sgc_version(), lib.sgc_version()
import numpy as np                               # noqa: E402
from sgcount_amd.workload import DeviceWorkload  # noqa: E402

ref = None
from sgcount_amd import synth                    # noqa: E402
DOM = synth.MODE_FIXED | synth.mode_dominant(60)  # a sample that one guide dominates: most workgroups of k_count_slices on one slice, segments
for n, ng, opts, mode in ((1_000_000, 20_000, {}, 0), (1_000_000, 20_000, {"balanced": 0}, 0), (1_000_000, 20_000, {"five_byte": 0}, 0),
                          (1_000_000, 20_000, {"five_byte": 0, "six_byte": 0}, 0),
                          (1_000_000, 20_000, {"direct": 0}, 0), (1_000_000, 20_000, {"dense": 0, "cuckoo": 0}, 0), (1_000_000, 20_000, {"variant": 3}, 0),
                          (40_000_000, 100_000, {}, 0), (40_000_000, 100_000, {"place_trials": 4}, 0), (3_000, 300, {}, 0), (1_500_000, 150_000, {}, 0),
                          (8_000_000, 100_000, {}, DOM), (8_000_000, 100_000, {"balanced": 0}, DOM), (2_000, 100_000, {}, DOM),
                          (8_000_000, 100_000, {}, 0), (8_000_000, 100_000, {"wide": 0}, 0),          # the narrow loop of k_count_slices against the wide one
                          (1_500_000, 250_000, {}, 0), (1_500_000, 250_000, {"cuckoo": 0}, 0)):       # slices of 2^13 slots
    for exact in (False, True):
        wl = DeviceWorkload(n, ng, 20, one_mismatch=not exact, gen_chunk=2_000_000, mode=mode)
        for k, v in opts.items():
            wl.dl.set_option(k, v)
        wl.step()
        counts, total, matched = wl.result()          # sgc_sample_export + sync: the flag word is read by finish below
        out = np.zeros(ng, dtype=np.uint64)
        import ctypes as C
        t, m = C.c_uint64(), C.c_uint64()
        _ffi.check(lib.sgc_sample_finish(wl.sample, out.ctypes.data, C.byref(t), C.byref(m)))       # raises on a bounds flag
        assert total == n and int(counts.sum()) == matched == m.value
        key = (n, ng, exact, mode)
        if not opts:
            ref = ref or {}
            ref[key] = counts.tolist()
        elif key in (ref or {}):
            assert counts.tolist() == ref[key], (n, ng, opts, exact)
        # split pushes through the same checked kernels
        wl.step(0, n // 3)
        a = wl.result()[0]
        wl.step(n // 3, n - n // 3)
        b = wl.result()[0]
        assert (a + b).tolist() == counts.tolist()
        wl.close()
        print("ok", n, ng, opts, "dominant" if mode else "", "exact" if exact else "1mm", flush=True)
print("CHECKED BUILD OK")
