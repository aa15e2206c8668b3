#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native sgRNA count path.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the
driver's launch line: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) and invoked directly —
`python bench.py --gpus N` then starts N fresh child processes itself, one per GPU, BEFORE anything in this process
has touched the GPU, waits for them and exits with their status.

A *step* is one pass of the hot path (per-read offset scan, library lookup, single-mismatch probe, count — reference
src/counter.rs:96-236) over one whole sample of packed reads already resident in HBM, ending with the u64 count
vector + totals exported on the device (one row per step).  For N > 1 every rank counts its own samples (seed + rank;
weak scaling, no data-path collective): a step is one sample, and the rows of all K samples of all ranks are exchanged
with ONE RCCL all-gather at the end of the batch, inside the timed region (fewer, larger collectives: K x 0.8 MB per
rank); the cost of that exchange is also measured on its own (`exchange`).  Prints ONE JSON line (rank 0).

Workload (BASELINE.json): 100k-guide synthetic library, 100M x 150 bp synthetic reads, guide at offset 30 (-a 30),
position recursion on; default `--workload 1mm` = configs[2] (the reference's default mode, exact + one mismatch — the
configuration the north-star target is quoted on); `--workload exact` = configs[1] (-x).

Besides the headline `value` (records resident in HBM), the line carries
  roofline      the count pipeline against the HBM roofline, on both byte accountings (18 B/read with the
                permute-table sector, and the 8 B/read the shipped path actually needs)
  cpu_baseline  the CPU port (oracle/, 1 thread like the reference within a sample) on a bounded prefix, plus an
                N-thread courtesy line and the probe for a reference binary
  e2e           end to end: synthetic FASTQ *text* of the same sample in the page cache -> `sgcount-hip` -> count
                table, plain and .gz, with the stages timed, next to the CPU port's end-to-end rate (N = 1 only)
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES = {"exact": 8.0, "1mm": 18.0}   # algorithmic bytes per read
NEEDED_BYTES = 8.0              # what the shipped path needs per read: one packed record (variant 4 reads no permute sector)
FASTQ_BYTES_PER_READ = 316.0    # text of one synthetic record


def kernel_sources_sha16():
    """identifies the device code the PMC traffic figures of profiles/pmc_traffic.json belong to"""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "sgcount_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "sgc_*.h"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per sample (per GPU)")
    ap.add_argument("--guides", type=int, default=100_000)
    ap.add_argument("--workload", choices=["1mm", "exact"], default="1mm")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-oracle budget; 0 disables the baseline")
    ap.add_argument("--variant", type=int, default=None, help="count-kernel variant (tuning)")
    ap.add_argument("--e2e-reads", type=int, default=100_000_000, help="reads of the end-to-end FASTQ leg; 0 disables it")
    ap.add_argument("--e2e-gz-reads", type=int, default=30_000_000, help="reads of the .gz end-to-end leg; 0 disables it")
    ap.add_argument("--e2e-dir", default=None, help="where the FASTQ text is written (default: /dev/shm or /tmp)")
    ap.add_argument("--dominant", type=int, default=40, help="percent of the reads of the skewed-sample leg that draw ONE guide; 0 disables the leg")
    ap.add_argument("--placement-trials", type=int, default=1, help="0: skip the second timed region that prices the opt-in placement trials (profiling runs)")
    ap.add_argument("--other-configs", type=int, default=1, help="0: skip the short legs on BASELINE.json configs[1] (exact) and configs[3] (auto-offset)")
    ap.add_argument("--multi-sample-reads", type=int, default=25_000_000, help="reads per sample of the multi-sample end-to-end leg (4 samples at N = 1, N at N > 1 through ONE command line); 0 disables it")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# N > 1 invoked directly: fresh child processes, one per rank, before this process touches the GPU
# ---------------------------------------------------------------------------------------------------------
def self_launch(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGC_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:            # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    sys.exit(rc)


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (bench.py's cpu_baseline leg is one of the three places that may load oracle/)
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(lib_seqs, L, offset, exact, recursion, seed, mode, budget_s):
    """Times the CPU oracle (a port of the reference's algorithm, 1 thread like the reference within a sample) on a
    bounded prefix of the same workload, FASTQ text in -> counts out; then the same port on N threads over disjoint
    chunks as a courtesy "best CPU" line (NOT how the reference runs one sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from sgcount_amd import synth
    lib_text = synth.library_fasta(lib_seqs)
    t0 = time.perf_counter()
    lib = O.Library(lib_text)
    perm = None if exact else O.Permuter(lib)
    setup_s = time.perf_counter() - t0
    ctr = O.Counter(lib, perm, False, offset, L, recursion)
    chunk, done, spent = 250_000, 0, 0.0
    while spent < budget_s and done < 50_000_000:
        text = synth.fastq_host(lib_seqs, done, chunk, seed, mode)     # generation is not timed
        t = time.perf_counter()
        ctr.feed_text(text)
        spent += time.perf_counter() - t
        done += chunk
    base = {"value": done / spent, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the same synthetic sample as FASTQ text in memory (%.1f s of CPU; one-time "
                      "library%s setup %.1f s excluded)" % (done, spent, "" if exact else "+permuter", setup_s),
            "setup_s": setup_s, "host_cpus": os.cpu_count(), "host_cpus_usable": len(os.sched_getaffinity(0)), "host_cpus_cgroup_quota": _cpu_quota(),
            # probe for the real thing — expected absent (no Rust toolchain, no crates mirror)
            "reference_binary": shutil.which("sgcount"), "cargo": shutil.which("cargo")}
    # courtesy line: the same port on N threads (ctypes releases the GIL), every thread its own Counter over its own chunks
    try:
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
        per_thread = 2
        texts = [[synth.fastq_host(lib_seqs, (t * per_thread + k) * chunk, chunk, seed, mode) for k in range(per_thread)]
                 for t in range(threads)]
        ctrs = [O.Counter(lib, perm, False, offset, L, recursion) for _ in range(threads)]

        def work(t):
            for x in texts[t]:
                ctrs[t].feed_text(x)
        t = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, range(threads)))
        dt = time.perf_counter() - t
        base["n_thread_courtesy"] = {"value": threads * per_thread * chunk / dt, "unit": "reads/s", "cores": threads,
                                     "note": "same port, reads dealt to N threads with private counters — not how the "
                                             "reference runs a sample (one rayon task per sample, src/count.rs:117)"}
    except Exception as e:                      # the courtesy line must never break the bench
        base["n_thread_courtesy"] = {"error": repr(e)}
    return base, ctr, done


# ---------------------------------------------------------------------------------------------------------
# end to end: FASTQ text in the page cache -> sgcount-hip -> table
# ---------------------------------------------------------------------------------------------------------
def _free_bytes(path):
    try:
        st = os.statvfs(path)
        return st.f_bavail * st.f_frsize
    except OSError:
        return 0


def _mem_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def _run_cli(cli, argv, stats=None):
    t0, u0 = time.perf_counter(), time.time()
    p = subprocess.run([cli] + argv + (["--stats-json", stats] if stats else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt, u1 = time.perf_counter() - t0, time.time()
    if p.returncode != 0:
        raise RuntimeError("sgcount-hip failed (%d): %s" % (p.returncode, p.stderr.decode()[-400:]))
    st = json.load(open(stats)) if stats else None
    if st and "count_entered_unix_s" in st:
        # what the stage timers inside count() do not see: exec + dynamic linking + argument handling before it, and
        # the teardown (device contexts, pinned buffers, process exit) after the table is written
        st["process_start_to_count_s"] = st["count_entered_unix_s"] - u0
        st["teardown_s"] = u1 - st["stats_written_unix_s"]
    return dt, st


def _table_counts(path, n_guides):
    import numpy as np
    counts = np.zeros(n_guides, dtype=np.uint64)
    with open(path, "rb") as f:
        next(f)
        for line in f:
            g, c = line.split(b"\t")
            counts[int(g[2:])] = int(c)
    return counts


def write_fastq(wl, n, path, chunk=2_000_000):
    """The first n reads of the workload's sample as FASTQ text (generated on the device, written through a pinned buffer)."""
    import torch
    from sgcount_amd import synth
    pinned = None
    with open(path, "wb", buffering=0) as f:
        for first in range(0, n, chunk):
            m = min(chunk, n - first)
            text, _ = synth.fastq_device(wl.lib_dev, first, m, wl.reads_seed, wl.mode)
            if pinned is None or pinned.numel() < text.numel():
                pinned = torch.empty(int(text.numel() * 1.05), dtype=torch.uint8, pin_memory=True)
            pinned[: text.numel()].copy_(text)
            torch.cuda.synchronize()
            f.write(memoryview(pinned.numpy())[: text.numel()])
            del text
    del pinned


def _cpu_quota():
    """CPUs the cgroup grants this process (cpu.max quota / period), or None: the affinity mask of a GPU box shows every core of the host"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                return None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            q = float(txt[0])
            return None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError, IndexError):
            continue
    return None


def _fractions(reads, steps, kernel_ms, algo_bytes):
    """both byte accountings of DESIGN.md §6 for one timed region"""
    a = algo_bytes * reads * steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
    n = NEEDED_BYTES * reads * steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
    return {"achieved_GBps": a, "frac": a / HBM_PEAK_GBPS if a else None, "algorithmic_bytes_per_read": algo_bytes,
            "achieved_on_needed_bytes_GBps": n, "frac_on_needed_bytes": n / HBM_PEAK_GBPS if n else None}


def _short_leg(wl, steps, warmup, algo_bytes):
    """`steps` passes over wl's resident sample, timed like the headline (wall clock around synchronised steps + the library's per-kernel events)"""
    import torch
    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize()
    wl.dl.timing(True)
    wl.dl.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = wl.dl.timing(reset=True)
    wl.dl.timing(False)
    kernel_ms = tm.part_ms + tm.lookup_ms + tm.miss_ms + tm.hist_ms
    counts, total, matched = wl.result()
    assert total == wl.n_reads and int(counts.sum()) == matched, "count-sum invariant violated"
    return {"steps": steps, "ms_per_step": 1e3 * el / steps, "value": wl.n_reads * steps / el, "unit": "reads/s",
            "kernel_ms_per_step": kernel_ms / steps,
            "kernels": {"partition_ms_per_step": tm.part_ms / steps, "slice_count_ms_per_step": tm.lookup_ms / steps,
                        "miss_resolve_ms_per_step": tm.miss_ms / steps, "histogram_ms_per_step": tm.hist_ms / steps},
            "roofline": _fractions(wl.n_reads, steps, kernel_ms, algo_bytes), "matched_fraction": matched / total}


def other_configs(args, exact_headline, L, offset, recursion, dev_index):
    """BASELINE.json's other 1-GPU configurations, each as a short leg of its own resident sample (5 steps, outside the headline's
    timed region): configs[1] -x (or configs[2] when the headline itself is -x) and configs[3], the variable-adapter sample whose
    offset the offsetter has to find first (src/main.rs:163-176 -> src/offsetter.rs:185-210) — its wall time on the 5000-read head
    of the sample's own FASTQ text is reported beside the pass."""
    import tempfile
    from sgcount_amd import hostlib, synth
    from sgcount_amd.workload import DeviceWorkload
    out = {}
    steps, warmup = 5, 2
    other_exact = not exact_headline
    wl = DeviceWorkload(args.reads, args.guides, L, one_mismatch=not other_exact, position_recursion=recursion, offset=offset,
                        reads_seed=synth.READS_SEED, device_index=dev_index)
    try:
        leg = _short_leg(wl, steps, warmup, ALGO_BYTES["exact" if other_exact else "1mm"])
        leg["workload"] = "BASELINE.json configs[%d]: same library and reads, %s" % (1 if other_exact else 2, "-x exact only" if other_exact else "exact + 1 mismatch")
        out["configs[%d]" % (1 if other_exact else 2)] = leg
    finally:
        wl.close()
    del wl
    # configs[3]: adapter length drawn from {28..32}; no -a: the offsetter reads the library and the first 5000 reads
    wl = DeviceWorkload(args.reads, args.guides, L, one_mismatch=not exact_headline, position_recursion=recursion, offset=offset,
                        reads_seed=synth.READS_SEED, device_index=dev_index, mode=synth.MODE_STAGGER)
    try:
        with tempfile.TemporaryDirectory() as d:
            lp, fp = os.path.join(d, "library.fa"), os.path.join(d, "head.fastq")
            open(lp, "wb").write(synth.library_fasta(wl.lib_seqs))
            open(fp, "wb").write(synth.fastq_host(wl.lib_seqs, 0, 6000, synth.READS_SEED, synth.MODE_STAGGER))
            hostlib.entropy_offset_group(lp, [fp], 5000)            # (first call: the library file enters the page cache)
            t0 = time.perf_counter()
            (rev, idx), = hostlib.entropy_offset_group(lp, [fp], 5000)
            off_s = time.perf_counter() - t0
        leg = _short_leg(wl, steps, warmup, ALGO_BYTES["exact" if exact_headline else "1mm"])
        leg["workload"] = ("BASELINE.json configs[3]: adapter of 28..32 bases (5/10/70/10/5 %), no -a; the records were packed at the offset the "
                           "offsetter must find; reads behind a 28- or 32-base adapter stay unmatched (src/counter.rs:96-140 tries +-1 only)")
        leg["offsetter"] = {"found": ("Reverse(%d)" if rev else "Forward(%d)") % idx, "expected": "Forward(%d)" % offset, "ok": (not rev) and idx == offset,
                            "wall_s": off_s, "reads_sampled": 5000, "note": "host f64 over the library (100k guides) + the 5000-read head: src/offsetter.rs:185-210"}
        out["configs[3]"] = leg
    finally:
        wl.close()
    return out



def multi_sample_leg(wl, args, exact, n_devices, where, want=None):
    """BASELINE.json configs[4] the way the reference runs it (src/main.rs:98-99,145, src/count.rs:117-136): S input files through
    ONE command line with -t S — here S = 4 samples on the one GPU at N = 1, S = N samples dealt to N devices at N > 1.  The files
    are consecutive quarters of the bench sample, so at N = 1 the columns must add up to the resident pass's table."""
    import numpy as np
    from sgcount_amd import hostlib, synth
    S = 4 if n_devices == 1 else n_devices
    per = int(args.multi_sample_reads)
    if per * S > args.reads:
        per = args.reads // S
    need = per * S * FASTQ_BYTES_PER_READ * 1.05
    avail = _mem_available()
    if _free_bytes(where) < need * 1.05 or (avail and avail < need * 1.5):
        return {"skipped": "no room for %d x %dM reads of FASTQ text under %s" % (S, per // 1_000_000, where)}
    d = os.path.join(where, "sgc_multi_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    try:
        lib_path, table, stats = os.path.join(d, "library.fa"), os.path.join(d, "table.tsv"), os.path.join(d, "stats.json")
        open(lib_path, "wb").write(synth.library_fasta(wl.lib_seqs))
        t0 = time.perf_counter()
        paths = []
        for k in range(S):
            fp = os.path.join(d, "s%d.fastq" % k)
            pinned_write_range(wl, k * per, per, fp)
            paths.append(fp)
        write_s = time.perf_counter() - t0
        cli = hostlib.cli_path()
        argv = ["-l", lib_path, "-i"] + paths + ["-a", "30", "-q", "-o", table, "-t", str(S), "--devices", str(n_devices)] + (["-x"] if exact else [])
        runs = [_run_cli(cli, argv, stats=stats) for _ in range(2)]
        wall, st = min(runs, key=lambda r: r[0])
        cols = np.zeros((S, args.guides), dtype=np.uint64)
        with open(table, "rb") as f:
            next(f)
            for line in f:
                cells = line.rstrip(b"\n").split(b"\t")
                cols[:, int(cells[0][2:])] = [int(c) for c in cells[1:]]
        smp = st["samples"]
        out = {"samples": S, "reads_per_sample": per, "devices": n_devices, "worker_threads": st.get("worker_threads"), "contexts": st.get("contexts"),
               "wall_s": wall, "wall_s_all_runs": [r[0] for r in runs], "reads_per_s": S * per / wall,
               "sample_s": [x["wall_s"] for x in smp], "sample_device": [x.get("device") for x in smp],
               "scanner_threads_per_sample": [x["reader_threads"] for x in smp],
               "scanner_busy_s_sum_over_threads": [x["read_busy_s"] for x in smp],
               "scanner_busy_s_all_samples": sum(x["read_busy_s"] for x in smp),
               "host_cpus_usable": len(os.sched_getaffinity(0)), "host_cpus_online": os.cpu_count(), "host_cpus_cgroup_quota": _cpu_quota(),
               "samples_s": st.get("samples_s"), "setup_s": st.get("setup_s"), "table_build_per_device_s": st.get("table_build_per_device_s"),
               "fastq_write_s": write_s, "reads_counted": [int(x["reads"]) for x in smp],
               "command": "sgcount-hip -l library.fa -i s0.fastq .. s%d.fastq -a 30 -t %d --devices %d" % (S - 1, S, n_devices),
               "note": "every sample is scanned by its own scanner threads (all samples share the host's usable CPUs: --scan-threads defaults to "
                       "min(16, cpus - 1) / worker threads) — the aggregate is bound by the host's scan of the text, not by the GPU(s)"}
        if want is not None and per * S == args.reads:
            out["columns_add_up_to_the_resident_pass"] = bool(np.array_equal(cols.sum(axis=0), want))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pinned_write_range(wl, first, n, path, chunk=2_000_000):
    """reads [first, first + n) of the workload's sample as FASTQ text"""
    import torch
    from sgcount_amd import synth
    pinned = None
    with open(path, "wb", buffering=0) as f:
        for a in range(first, first + n, chunk):
            m = min(chunk, first + n - a)
            text, _ = synth.fastq_device(wl.lib_dev, a, m, wl.reads_seed, wl.mode)
            if pinned is None or pinned.numel() < text.numel():
                pinned = torch.empty(int(text.numel() * 1.05), dtype=torch.uint8, pin_memory=True)
            pinned[: text.numel()].copy_(text)
            torch.cuda.synchronize()
            f.write(memoryview(pinned.numpy())[: text.numel()])
            del text
    del pinned


def e2e_block(wl, args, exact, cpu):
    """Synthetic FASTQ text of the bench sample -> sgcount-hip (process start to finished table)."""
    import numpy as np
    import torch
    from sgcount_amd import hostlib, synth
    out = {}
    n = min(args.e2e_reads, args.reads)
    need = n * FASTQ_BYTES_PER_READ * 1.02
    cands = [args.e2e_dir] if args.e2e_dir else ["/dev/shm", "/tmp"]
    avail = _mem_available()
    where = None
    for c in cands:
        if c and os.path.isdir(c) and _free_bytes(c) > need * 1.05 and (avail == 0 or avail > need * 1.5):
            where = c
            break
    if where is None:      # scale the sample down to what fits
        best = max(cands, key=lambda c: _free_bytes(c) if c and os.path.isdir(c) else 0)
        room = min(_free_bytes(best), avail / 1.5 if avail else 1e18)
        n = int(min(n, room / (FASTQ_BYTES_PER_READ * 1.1)) // 1_000_000 * 1_000_000)
        where = best
        if n < 1_000_000:
            return {"skipped": "no room for FASTQ text under %s" % cands}
    d = os.path.join(where, "sgc_e2e_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    try:
        lib_path, fq, table = os.path.join(d, "library.fa"), os.path.join(d, "reads.fastq"), os.path.join(d, "table.tsv")
        open(lib_path, "wb").write(synth.library_fasta(wl.lib_seqs))
        t0 = time.perf_counter()
        chunk = 2_000_000
        write_fastq(wl, n, fq, chunk)
        size = os.path.getsize(fq)
        out["fastq"] = {"reads": n, "bytes": size, "dir": where, "write_s": time.perf_counter() - t0}
        cli = hostlib.cli_path()
        base = ["-l", lib_path, "-a", "30", "-q", "-o", table] + (["-x"] if exact else [])
        # plain text: three runs, every one with the stage timers on.  The FIRST run of a fresh box is reported on its own
        # (wall_s_first_run): it is the run a user makes, and it pays for whatever is cold (shared libraries of the ROCm
        # runtime paged in from the image, first use of the device by a new process); wall_s is the best of the later runs.
        def stages_of(stats):
            smp = stats["samples"][0]
            return {"process_start_to_count_s": stats.get("process_start_to_count_s"), "library_load_s": stats["library_load_s"],
                    "hip_runtime_wait_s": stats.get("hip_runtime_wait_s"), "device_init_s": stats.get("device_init_s"), "table_build_s": stats["table_build_s"],
                    "sample_s": smp["wall_s"], "table_write_s": stats["table_write_s"], "context_free_s": stats.get("context_free_s"),
                    "process_exit_s": stats.get("teardown_s"),
                    "feeder_setup_s": smp.get("feeder_setup_s"), "first_push_s": smp.get("first_push_s"),
                    "path": "host scan -> packed records" if smp.get("scan_path") else ("FASTQ text parsed on the GPU" if smp.get("text_path") else "record reader"),
                    "host_copy_s": smp.get("host_copy_s"),
                    "file_read_busy_s_sum_over_threads": smp["read_busy_s"], "reader_threads": smp["reader_threads"],
                    "host_waited_for_text_s": smp["wait_for_text_s"], "host_waited_for_upload_s": smp["wait_for_upload_s"],
                    "host_in_push_calls_s": smp["push_s"], "drain_s": smp["finish_s"],
                    "h2d_s": smp["h2d_ms"] / 1e3, "ingest_kernels_s": smp["ingest_kernels_ms"] / 1e3,
                    "count_kernels_s": smp["count_kernels_ms"] / 1e3}
        runs = [_run_cli(cli, base + ["-i", fq], stats=os.path.join(d, "stats%d.json" % k)) for k in range(3)]
        got = _table_counts(table, args.guides)
        wl.step(0, n)
        want, total, matched = wl.result()
        parity = bool(np.array_equal(got, want))
        walls = [r[0] for r in runs]
        best = min(range(1, len(runs)), key=lambda k: walls[k])
        wall, stats = walls[best], runs[best][1]
        smp = stats["samples"][0]
        out["plain"] = {
            "wall_s": wall, "reads_per_s": n / wall, "text_GBps": size / wall / 1e9, "wall_s_all_runs": walls,
            "wall_s_first_run": walls[0], "reads_per_s_first_run": n / walls[0],
            "table_equals_resident_pass": parity,
            "stages": dict(stages_of(stats), note="stage timers of the best later run (HIP events on in every run); uploads, ingest "
                           "and count kernels of consecutive parts overlap, so the stages do not add up to sample_s"),
            "stages_first_run": stages_of(runs[0][1]),
        }
        # the validated fallback (and the path of .gz / BGZF input): the text itself goes over PCIe and the GPU parses it
        # (two runs, the better one: the first pread() of the file pays the LRU activation the scan path avoids — DESIGN.md §5)
        text_runs = [_run_cli(cli, base + ["-i", fq, "--pack", "fastq"], stats=os.path.join(d, "stats_text.json")) for _ in range(2)]
        wall_t, stats_t = min(text_runs, key=lambda r: r[0])
        smp_t = stats_t["samples"][0]
        out["plain_gpu_parsed_text"] = {
            "wall_s": wall_t, "wall_s_all_runs": [r[0] for r in text_runs], "reads_per_s": n / wall_t, "table_equals_resident_pass": bool(np.array_equal(_table_counts(table, args.guides), want)),
            "stages": stages_of(stats_t),
            "ingest_text_GBps_vs_hbm": {"achieved": size / max(smp_t["ingest_kernels_ms"], 1e-9) / 1e6, "peak": HBM_PEAK_GBPS,
                                         "note": "FASTQ text bytes / Σ(k_fastq_count + k_scan_tiles + k_fastq_pack) time"},
        }
        if cpu:
            rate, setup = cpu["value"], cpu["setup_s"]
            out["cpu_port"] = {"reads_per_s_setup_excluded": rate, "reads_per_s_setup_included": n / (n / rate + setup),
                               "setup_s": setup, "cores": 1,
                               "note": "the oracle's measured FASTQ-text rate (cpu_baseline) projected to the e2e sample size"}
            inc = out["cpu_port"]["reads_per_s_setup_included"]
            out["speedup_plain"] = {"vs_cpu_setup_excluded": out["plain"]["reads_per_s"] / rate,
                                    "vs_cpu_setup_included": out["plain"]["reads_per_s"] / inc,
                                    "first_run_vs_cpu_setup_excluded": out["plain"]["reads_per_s_first_run"] / rate,
                                    "first_run_vs_cpu_setup_included": out["plain"]["reads_per_s_first_run"] / inc}
        # .gz: one deflate stream — inflated by several threads (speculative chunk decoding, sgh_inflate.cpp); the CPU line next to
        # it inflates on one thread like the reference (flate2 inside fxread on the sample's thread)
        ngz = min(args.e2e_gz_reads, n)
        if ngz > 0:
            t0 = time.perf_counter()
            gz = os.path.join(d, "reads.fastq.gz")
            # the first ngz reads as their own text file, compressed as 16 gzip members in parallel (zlib reads
            # concatenated members as one stream)
            src = fq
            if ngz < n:
                src = os.path.join(d, "head.fastq")
                with open(src, "wb", buffering=0) as f:
                    for first in range(0, ngz, chunk):
                        m = min(chunk, ngz - first)
                        text, _ = synth.fastq_device(wl.lib_dev, first, m, wl.reads_seed, wl.mode)
                        f.write(memoryview(text.cpu().numpy()))
                        del text
            gz_bytes = os.path.getsize(src)
            parts = 16
            per = (gz_bytes + parts - 1) // parts
            procs = []
            for k in range(parts):
                lo, hi = k * per, min(gz_bytes, (k + 1) * per)
                if lo >= hi:
                    break
                cmd = "tail -c +%d %s | head -c %d | gzip -1 > %s.%02d" % (lo + 1, src, hi - lo, gz, k)
                procs.append(subprocess.Popen(["bash", "-c", cmd]))
            for p in procs:
                if p.wait() != 0:
                    raise RuntimeError("gzip failed")
            with open(gz, "wb") as o:
                for k in range(len(procs)):
                    with open("%s.%02d" % (gz, k), "rb") as part:
                        shutil.copyfileobj(part, o, 1 << 24)
                    os.remove("%s.%02d" % (gz, k))
            prep = time.perf_counter() - t0
            gz_runs = [_run_cli(cli, base + ["-i", gz], stats=os.path.join(d, "stats_gz.json")) for _ in range(2)]
            wall_gz, stats = min(gz_runs, key=lambda r: r[0])
            got = _table_counts(table, args.guides)
            wl.step(0, ngz)
            want_gz, total_gz, _ = wl.result()
            smp = stats["samples"][0]
            # one-thread inflate rate of the same file (zlib, like the reference's flate2 on its single sample thread)
            import zlib
            t0 = time.perf_counter()
            dec, inflated = zlib.decompressobj(31), 0
            with open(gz, "rb") as f:
                while True:
                    b = f.read(1 << 22)
                    if not b:
                        break
                    while b:
                        inflated += len(dec.decompress(b))
                        b = dec.unused_data
                        if dec.eof:
                            dec = zlib.decompressobj(31)
                        else:
                            break
            inflate_s = time.perf_counter() - t0
            out["gz"] = {"reads": int(smp["reads"]), "gz_bytes": os.path.getsize(gz), "wall_s": wall_gz,
                         "reads_per_s": smp["reads"] / wall_gz, "table_equals_resident_pass": bool(np.array_equal(got, want_gz)) and
                         int(smp["reads"]) == total_gz, "sample_s": smp["wall_s"], "inflate_busy_s": smp["read_busy_s"],
                         "zlib_inflate_alone_s": inflate_s, "prepare_s": prep, "wall_s_all_runs": [r[0] for r in gz_runs],
                         "parallel_gzip": bool(smp.get("parallel_gzip")), "inflate_threads": smp["reader_threads"],
                         "gzip_chunks_decoded_in_order": smp.get("gzip_chunks_decoded_in_order"),
                         "stages": stages_of(stats),
                         "note": "ONE deflate stream (16 members written by 16 gzip processes, no BGZF size fields): zlib on one core inflates "
                                 "it at ~%.2f GB/s of text; here its chunks are decoded speculatively by %d threads, stitched in order and packed into records by "
                                 "the same threads (the host scan: nothing of it needs the device, so it runs while the device starts up and the tables are built)"
                                 % (inflated / inflate_s / 1e9, smp["reader_threads"])}
            if cpu:
                r = smp["reads"]
                cpu_gz = r / (inflate_s + r / cpu["value"])
                out["gz"]["cpu_port_reads_per_s"] = cpu_gz
                out["gz"]["speedup_vs_cpu"] = out["gz"]["reads_per_s"] / cpu_gz
                out["gz"]["cpu_note"] = "CPU port composed as inflate (measured here, zlib) + count (cpu_baseline rate) on one thread"
            # BGZF (bgzip / htslib / Illumina converters): the same text as gzip members of <= 64 KiB, which the host inflates
            # with all its reader threads
            t0 = time.perf_counter()
            bg = os.path.join(d, "reads.bgzf.fastq.gz")
            procs = []
            for k in range(parts):
                lo, hi = k * per, min(gz_bytes, (k + 1) * per)
                if lo >= hi:
                    break
                procs.append(subprocess.Popen([sys.executable, "-m", "sgcount_amd.bgzf", src, str(lo), str(hi), "%s.%02d" % (bg, k)],
                                              cwd=os.path.dirname(os.path.abspath(__file__))))
            for p in procs:
                if p.wait() != 0:
                    raise RuntimeError("bgzf compression failed")
            from sgcount_amd.bgzf import EOF_MARKER
            with open(bg, "wb") as o:
                for k in range(len(procs)):
                    with open("%s.%02d" % (bg, k), "rb") as part:
                        shutil.copyfileobj(part, o, 1 << 24)
                    os.remove("%s.%02d" % (bg, k))
                o.write(EOF_MARKER)
            prep = time.perf_counter() - t0
            walls = []
            for _ in range(2):
                w, stats = _run_cli(cli, base + ["-i", bg], stats=os.path.join(d, "stats_bgzf.json"))
                walls.append(w)
            got = _table_counts(table, args.guides)
            smp = stats["samples"][0]
            out["bgzf"] = {"reads": int(smp["reads"]), "gz_bytes": os.path.getsize(bg), "wall_s": min(walls), "wall_s_all_runs": walls,
                           "reads_per_s": smp["reads"] / min(walls),
                           "table_equals_resident_pass": bool(np.array_equal(got, want_gz)) and int(smp["reads"]) == total_gz,
                           "sample_s": smp["wall_s"], "inflate_busy_s_sum_over_threads": smp["read_busy_s"],
                           "reader_threads": smp["reader_threads"], "prepare_s": prep,
                           "stages": stages_of(stats),
                           "note": "BGZF members are independent deflate streams: runs of whole members are inflated (zlib) and packed into records by the "
                                   "scanner's threads, beside the device's start-up like plain gzip; --pack fastq inflates them into pinned slices for the GPU parser"}
            if cpu and "cpu_port_reads_per_s" in out.get("gz", {}):
                out["bgzf"]["speedup_vs_cpu"] = out["bgzf"]["reads_per_s"] / out["gz"]["cpu_port_reads_per_s"]
            # libraries with guides outside ACGT (upstream compares raw bytes: library.rs:89-99): the same text against the same
            # library with an 'N' put into 100 of its guides (hybrid: packed pass + byte-string chain for the reads near those
            # guides) and into 60 % of them (byte-string path alone), next to the all-ACGT library on the same GPU-parsed-text
            # path.  Timing only: the tables differ from the ACGT library's by construction; parity is tests/test_generic_gpu.py.
            seqs = wl.lib_seqs.copy()
            libs = {}
            for name, every in (("hybrid_100_N_guides", len(seqs) // 100), ("bytes_60pct_N_guides", 0)):
                s2 = seqs.copy()
                # (not among the last 200 guides: the planted Hamming-1 pairs there could become duplicates)
                idx = np.arange(0, len(seqs) - 200, every) if every else np.arange(len(seqs) - 200)[np.arange(len(seqs) - 200) % 5 < 3]
                s2[idx, 7] = ord("N")
                lp = os.path.join(d, name + ".fa")
                open(lp, "wb").write(synth.library_fasta(s2))
                libs[name] = lp
            libs["acgt"] = lib_path
            legs = {}
            for name, lp in libs.items():
                b2 = ["-l", lp, "-a", "30", "-q", "-o", table] + (["-x"] if exact else [])
                for mode, extra in (("default", []), ("gpu_parsed_text", ["--pack", "fastq"])):
                    rr = [_run_cli(cli, b2 + ["-i", src] + extra, stats=os.path.join(d, "stats_lib.json")) for _ in range(2)]
                    w, st2 = min(rr, key=lambda r: r[0])
                    s0 = st2["samples"][0]
                    legs.setdefault(name, {})[mode] = {
                        "wall_s": w, "reads_per_s": s0["reads"] / w, "sample_s": s0["wall_s"], "table_build_s": st2["table_build_s"],
                        "count_kernels_s": s0["count_kernels_ms"] / 1e3, "ingest_kernels_s": s0["ingest_kernels_ms"] / 1e3,
                        "path": "host scan -> packed records (+ bytes of the routed reads)" if s0.get("scan_path") else "FASTQ text parsed on the GPU"}
            # ... and the hybrid library on the whole sample, where the one-time table build hides behind the scan of the text
            b2 = ["-l", libs["hybrid_100_N_guides"], "-a", "30", "-q", "-o", table] + (["-x"] if exact else [])
            rr = [_run_cli(cli, b2 + ["-i", fq], stats=os.path.join(d, "stats_lib.json")) for _ in range(2)]
            w, st2 = min(rr, key=lambda r: r[0])
            legs["hybrid_100_N_guides"]["whole_sample"] = {"reads": int(st2["samples"][0]["reads"]), "wall_s": w, "reads_per_s": st2["samples"][0]["reads"] / w,
                                                           "vs_acgt_library": out["plain"]["wall_s"] / w, "table_build_s": st2["table_build_s"],
                                                           "sample_s": st2["samples"][0]["wall_s"]}
            out["non_acgt_libraries"] = dict(legs, reads=int(ngz),
                                             hybrid_vs_acgt=legs["hybrid_100_N_guides"]["default"]["reads_per_s"] / legs["acgt"]["default"]["reads_per_s"],
                                             hybrid_vs_acgt_sample_time=legs["acgt"]["default"]["sample_s"] / legs["hybrid_100_N_guides"]["default"]["sample_s"],
                                             bytes_vs_acgt_gpu_parsed_text=legs["bytes_60pct_N_guides"]["gpu_parsed_text"]["reads_per_s"] / legs["acgt"]["gpu_parsed_text"]["reads_per_s"],
                                             note="default = what the command line picks (plain text: the host scan; a library of mostly non-ACGT guides: "
                                                  "the byte-string path on GPU-parsed text); best of two runs each; the wall time of these runs is partly "
                                                  "process start-up and the one-time table build")
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if args.multi_sample_reads > 0:
        try:                                      # (the single sample's text is gone by now: the four files take its place)
            wl.step()
            want_all, _, _ = wl.result()
            out["multi_sample"] = multi_sample_leg(wl, args, exact, 1, where, want_all)
        except Exception as e:
            out["multi_sample"] = {"error": repr(e)[:500]}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                      # never returns

    import torch
    import torch.distributed as dist
    from sgcount_amd import synth
    from sgcount_amd.workload import DeviceWorkload
    from sgcount_amd.distributed import all_gather_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)          # one rank per GPU on a full node; ranks share a card only in rehearsals
    torch.cuda.set_device(dev_index)
    backend = os.environ.get("SGC_BENCH_BACKEND", "nccl")   # "gloo": rehearse the N > 1 flow on a box with fewer GPUs
    cpu_group = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
            # a host-side group for the one wait that must not occupy the devices: while rank 0 runs the multi-sample command line over
            # all N devices the other ranks wait here, not inside a device-side barrier kernel
            cpu_group = dist.new_group(backend="gloo")
        else:
            dist.init_process_group(backend=backend)

    L, offset, recursion = 20, 30, True
    exact = args.workload == "exact"
    wl = DeviceWorkload(args.reads, args.guides, L, one_mismatch=not exact, position_recursion=recursion, offset=offset,
                        reads_seed=synth.READS_SEED + rank, device_index=dev_index)
    if args.variant is not None:
        wl.dl.set_option("variant", args.variant)
    coll_dev = wl.dev if backend == "nccl" else torch.device("cpu")
    row_len = args.guides + 2
    # one row per sample (= step) of this rank; N > 1: the [world, K x row] matrix of the whole batch
    rows = torch.zeros((max(args.steps, 1), row_len), dtype=torch.int64, device=wl.dev)
    matrix = torch.zeros((world, rows.numel()), dtype=torch.int64, device=coll_dev) if world > 1 else None

    def step(i):
        wl.step(out=rows[i % rows.shape[0]])

    def exchange():            # one all-gather of every sample row of the batch over RCCL/xGMI (gloo rehearsal: via the host)
        if world > 1:
            all_gather_rows((rows if backend == "nccl" else rows.cpu()).view(-1), matrix)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    exchange()
    fence()
    wl.dl.timing(True)
    wl.dl.timing(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    exchange()
    fence()
    elapsed = time.perf_counter() - t0
    tm = wl.dl.timing(reset=True)
    wl.dl.timing(False)
    exch = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank holds every sample row of the batch: rank r's step-0 row must be what rank r counted
        got = matrix.view(world, rows.shape[0], row_len)[rank, 0].to(rows.device)
        assert torch.equal(got, rows[0]), "exchanged count matrix does not hold this rank's row"
        # the exchange on its own (outside the timed region): the whole batch of K rows, and a single sample row
        one = torch.zeros((world, row_len), dtype=torch.int64, device=coll_dev)

        def timed_exchange(fn, reps=5):
            fence()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            fence()
            x = torch.tensor([(time.perf_counter() - t) / reps], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            return float(x.item()) * 1e3
        batch_ms = timed_exchange(exchange)
        row_ms = timed_exchange(lambda: all_gather_rows((rows[0] if backend == "nccl" else rows[0].cpu()), one))
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index)})
        exch = {"backend": backend, "ranks_seen": dist.get_world_size(), "rank_devices": devs,
                "all_gather_batch_ms": batch_ms, "batch_rows_per_rank": rows.shape[0], "batch_bytes_per_rank": rows.numel() * 8,
                "all_gather_one_sample_ms": row_ms, "sample_row_bytes": row_len * 8,
                "note": "the timed region holds ONE all-gather per batch of K steps; all_gather_one_sample_ms is what a "
                        "per-sample exchange would cost per step instead"}

    counts, total, matched = wl.result(rows[(args.steps - 1) % rows.shape[0]])
    assert total == args.reads and int(counts.sum()) == matched, "count-sum invariant violated"

    # Placement trials (sgc_set_option "place_trials", OFF by default in the library and in `value` above): a host that
    # re-counts resident samples on one ctx — this loop is one — may let the first large pass try several placements of its
    # block pool and keep the fastest.  Priced separately: the same K steps again with the trials on, their one-off cost,
    # and the memory they held meanwhile.
    placement = None
    if args.reads >= (1 << 25) and args.placement_trials:
        import ctypes as C
        wl.dl.set_option("place_trials", 32)
        fence()
        t1 = time.perf_counter()
        step(0)                                   # this pass runs the search
        fence()
        first_ms = (time.perf_counter() - t1) * 1e3
        for i in range(max(args.warmup - 1, 0)):
            step(i)
        fence()
        wl.dl.timing(True)
        wl.dl.timing(reset=True)
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        fence()
        el2 = time.perf_counter() - t1
        tm2 = wl.dl.timing(reset=True)
        wl.dl.timing(False)
        if world > 1:
            tmax = torch.tensor([el2], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el2 = float(tmax.item())
        info = (C.c_uint64 * 4)()
        wl.abi.sgc_placement_info(wl.dl.ctx, info)
        c2, t2, m2 = wl.result(rows[(args.steps - 1) % rows.shape[0]])
        assert t2 == total and m2 == matched and bool((c2 == counts).all()), "placement trials changed the table"
        placement = {"value_with_placement_trials": world * args.reads * args.steps / el2, "ms_per_step": 1e3 * el2 / args.steps,
                     "kernel_ms_per_step": (tm2.part_ms + tm2.lookup_ms + tm2.miss_ms + tm2.hist_ms) / args.steps,
                     "search": {"candidates": int(info[0]), "transient_bytes": int(info[1]), "search_ms": int(info[2]) / 1e3,
                                "kept": int(info[3]), "first_pass_wall_ms": first_ms},
                     "note": "not part of `value`: the exchange is left out of this second timed region; the search is paid once per ctx "
                             "and is worth it only to a host that re-counts resident samples of >= 32M records"}
        wl.dl.set_option("place_trials", 1)

    out = {
        "metric": "reads/s", "value": world * args.reads * args.steps / elapsed, "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[%d]: %dk-guide synthetic library (seed 0x5EED0001), %dM x 150 bp "
                                "synthetic reads per GPU (seed 0x5EED0002+rank), -a 30, %s, position recursion on; packed "
                                "8-byte records resident in HBM") % (1 if exact else 2, args.guides // 1000,
                                                                     args.reads // 1_000_000,
                                                                     "-x exact only" if exact else "exact + 1 mismatch (reference default)"),
                   "reads_per_gpu": args.reads, "guides": args.guides, "guide_len": L, "record_bytes": wl.dl.record_bytes,
                   "parallelism": "1 sample per GPU per step" + ("" if world == 1 else ", one RCCL all-gather of all sample rows per batch of K steps")},
        "matched_fraction": matched / total,
        "launch": "self-launched children" if os.environ.get("SGC_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "single process"),
    }
    if exch:
        out["exchange"] = exch
    if placement:
        out["placement_trials"] = placement
    if world == 1 and args.dominant and args.reads >= (1 << 25):
        # A sample that one guide dominates (synth.mode_dominant: a screen after strong selection): with the same number of workgroups
        # for every library slice the slice of that guide decides the kernel's time; the shipped pass deals ALL slice blocks out in
        # equal shares (`balanced`).  Same steps, its own resident sample; not part of `value`.
        wl2 = DeviceWorkload(args.reads, args.guides, L, one_mismatch=not exact, position_recursion=recursion, offset=offset,
                             reads_seed=synth.READS_SEED, device_index=dev_index, mode=synth.MODE_FIXED | synth.mode_dominant(args.dominant))
        skew = {"percent_of_reads_on_one_guide": args.dominant, "reads": args.reads}
        tables = []
        for name, bal in (("balanced_shares", 1), ("equal_workgroups_per_slice", 0)):
            wl2.dl.set_option("balanced", bal)
            for _ in range(2):
                wl2.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                wl2.step()
            torch.cuda.synchronize()
            skew[name] = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / args.steps}
            skew[name]["reads_per_s"] = args.reads / (skew[name]["ms_per_step"] * 1e-3)
            c2, t2, m2 = wl2.result()
            tables.append((c2.tolist(), t2, m2))
        skew["same_table"] = tables[0] == tables[1]
        skew["top_guide_share"] = max(tables[0][0]) / max(tables[0][1], 1)
        skew["note"] = "shipped: balanced_shares; the other line is sgc_set_option(balanced, 0), the round-2 scheme"
        wl2.close()
        out["skewed_sample"] = skew
    if rank == 0:
        reads_timed = args.reads * args.steps
        # the count path is a short pipeline of kernels over the same reads; the roofline is
        # quoted for the pipeline as a whole: algorithmic bytes of the pass / Σ kernel time of the pass
        parts = {"partition": tm.part_ms, "slice_count": tm.lookup_ms, "miss_resolve": tm.miss_ms, "histogram": tm.hist_ms}
        kernels = {k + "_ms_per_step": v / args.steps for k, v in parts.items()}
        kernels["launches_per_step"] = tm.launches / args.steps
        dom_ms = sum(parts.values())
        bpr = ALGO_BYTES[args.workload]
        achieved = bpr * reads_timed / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None
        needed = NEEDED_BYTES * reads_timed / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None
        # HBM bytes per pass from the rocprofv3 PMC passes committed under profiles/ (tools/pmc_traffic.py:
        # separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command, gfx950 FETCH_SIZE correction calibrated
        # on k_partition's known byte count); only quoted for the configuration it was collected on
        traffic, traffic_note = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and args.reads == 100_000_000 and args.guides == 100_000:
            try:
                doc = json.load(open(pmc))
                d = doc.get(args.workload)
                # the counters are replayed from a committed collection (tools/evidence.sh), not measured by this run: quoted only while the
                # kernels are the ones they were collected on (sha256 over sgcount_amd/csrc/*.hip + sgc_*.h, recorded by tools/pmc_traffic.py)
                if doc.get("kernel_sources_sha16") == kernel_sources_sha16():
                    traffic = d["hbm_bytes_per_step"]
                    traffic_note = "%.1f HBM B/read measured (%s) vs %.0f algorithmic" % (d["bytes_per_read"], d.get("collected", "profiles/"), ALGO_BYTES[args.workload])
                else:
                    traffic_note = ("not quoted: profiles/pmc_traffic.json was collected on other kernel sources (%s, now %s) — re-run tools/evidence.sh"
                                    % (doc.get("kernel_sources_sha16"), kernel_sources_sha16()))
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                           "traffic_note": traffic_note,
                           "kernel": ("count pipeline (k_partition + k_count_slices + k_core<A> + k_core<B>)" if not exact
                                      else "count pipeline (k_partition + k_count_slices + k_core<exact>)"),
                           "algorithmic_bytes_per_read": bpr,
                           # the strict accounting: the shipped path reads no permute-table sector, a read is one 8-byte record
                           "needed_bytes_per_read": NEEDED_BYTES, "achieved_on_needed_bytes": needed,
                           "frac_on_needed_bytes": (needed / HBM_PEAK_GBPS) if needed else None,
                           "kernel_ms_per_step": dom_ms / args.steps, "kernels": kernels}
        base = None
        if world == 1 and args.cpu_seconds > 0:
            base, ctr, m = cpu_baseline(wl.lib_seqs, L, offset, exact, recursion, synth.READS_SEED, 0, args.cpu_seconds)
            out["cpu_baseline"] = base
            # free parity check: the GPU path on the very same prefix must give the oracle's table
            wl.step(0, min(m, args.reads))
            g_counts, g_total, g_matched = wl.result()
            if m <= args.reads:
                ok = (g_counts.tolist() == ctr.table() and g_matched == ctr.matched_reads() and g_total == ctr.total_reads())
                out["parity_vs_oracle_on_cpu_sample"] = "bit-exact" if ok else "MISMATCH"
            out["kernel_vs_cpu_note"] = ("value / cpu_baseline.value compares resident packed records on the GPU with FASTQ text on one "
                                         "CPU thread: not a like-for-like ratio — see e2e for file-to-table against the same CPU port")
        if world == 1 and args.e2e_reads > 0:
            try:
                out["e2e"] = e2e_block(wl, args, exact, base)
            except Exception as e:              # the e2e leg must never cost the headline line
                out["e2e"] = {"error": repr(e)[:500]}
        if world == 1 and args.other_configs and args.reads >= (1 << 25):
            try:
                out["other_configs"] = other_configs(args, exact, L, offset, recursion, dev_index)
            except Exception as e:
                out["other_configs"] = {"error": repr(e)[:500]}
        if world > 1 and args.multi_sample_reads > 0:
            # N samples through ONE command line dealt to the N devices (the other ranks wait at the barrier below, their devices idle):
            # the host-bound end-to-end curve next to the resident-record one
            try:
                cands = [args.e2e_dir] if args.e2e_dir else ["/dev/shm", "/tmp"]
                where = max(cands, key=lambda c: _free_bytes(c) if c and os.path.isdir(c) else 0)
                out["e2e"] = {"multi_sample": multi_sample_leg(wl, args, exact, world, where)}
            except Exception as e:
                out["e2e"] = {"multi_sample": {"error": repr(e)[:500]}}
        print(json.dumps(out), flush=True)
    wl.close()
    if world > 1:
        if cpu_group is not None:
            dist.barrier(group=cpu_group)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
