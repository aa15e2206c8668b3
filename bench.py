#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native sgRNA count path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* is one pass of the hot path (per-read offset scan, library lookup, single-mismatch probe,
count — reference src/counter.rs:96-236) over one whole sample of packed reads already resident in
HBM, ending with the u64 count vector + totals exported on the device (one row per step).  For N > 1 every
rank counts its own samples (seed + rank; weak scaling, no data-path collective): a step is one sample, and
the rows of all K samples of all ranks are exchanged with ONE RCCL all-gather at the end of the batch, inside
the timed region (fewer, larger collectives: K x 0.8 MB per rank).  Prints ONE JSON line (rank 0).

Workload (BASELINE.json): 100k-guide synthetic library, 100M x 150 bp synthetic reads, guide at
offset 30 (-a 30), position recursion on; default `--workload 1mm` = configs[2] (the reference's
default mode, exact + one mismatch — the configuration the north-star target is quoted on);
`--workload exact` = configs[1] (-x).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES = {"exact": 8.0, "1mm": 18.0}   # algorithmic bytes per read


def cpu_baseline(lib_seqs, n_guides, L, offset, exact, recursion, seed, mode, budget_s, gpu_prefix_counts=None):
    """Times the CPU oracle (a port of the reference's algorithm, 1 thread like the reference within a
    sample) on a bounded prefix of the same workload, FASTQ text in → counts out."""
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
    return {"value": done / spent, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the same synthetic sample as FASTQ text (%.1f s of CPU; one-time "
                      "library%s setup %.1f s excluded)" % (done, spent, "" if exact else "+permuter", setup_s),
            "host_cpus": os.cpu_count()}, ctr, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per sample (per GPU)")
    ap.add_argument("--guides", type=int, default=100_000)
    ap.add_argument("--workload", choices=["1mm", "exact"], default="1mm")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-oracle budget; 0 disables the baseline")
    ap.add_argument("--variant", type=int, default=None, help="count-kernel variant (tuning)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sgcount_amd import synth
    from sgcount_amd.workload import DeviceWorkload
    from sgcount_amd.distributed import all_gather_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)          # one rank per GPU on a full node; ranks share a card only in rehearsals
    torch.cuda.set_device(dev_index)
    backend = os.environ.get("SGC_BENCH_BACKEND", "nccl")   # "gloo": rehearse the N > 1 flow on a box with fewer GPUs
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
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
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank holds every sample row of the batch: rank r's step-0 row must be what rank r counted
        got = matrix.view(world, rows.shape[0], row_len)[rank, 0].to(rows.device)
        assert torch.equal(got, rows[0]), "exchanged count matrix does not hold this rank's row"

    counts, total, matched = wl.result(rows[(args.steps - 1) % rows.shape[0]])
    assert total == args.reads and int(counts.sum()) == matched, "count-sum invariant violated"

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
    }
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
        # HBM bytes per pass from the rocprofv3 PMC passes committed under profiles/ (tools/pmc_traffic.py:
        # separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command, gfx950 FETCH_SIZE correction calibrated
        # on k_partition's known byte count); only quoted for the configuration it was collected on
        traffic, traffic_note = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and args.reads == 100_000_000 and args.guides == 100_000:
            try:
                d = json.load(open(pmc)).get(args.workload)
                traffic = d["hbm_bytes_per_step"]
                traffic_note = "%.1f HBM B/read measured vs %.0f algorithmic" % (d["bytes_per_read"], ALGO_BYTES[args.workload])
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                           "traffic_note": traffic_note,
                           "kernel": ("count pipeline (k_partition + k_count_slices + k_cp_count/k_cp_scatter + k_core, twice)" if not exact
                                      else "count pipeline (k_partition + k_count_slices + k_generic + k_resolve_miss + k_hist_segments)"),
                           "algorithmic_bytes_per_read": bpr,
                           "kernel_ms_per_step": dom_ms / args.steps, "kernels": kernels}
        if world == 1 and args.cpu_seconds > 0:
            base, ctr, m = cpu_baseline(wl.lib_seqs, args.guides, L, offset, exact, recursion, synth.READS_SEED, 0,
                                        args.cpu_seconds)
            out["cpu_baseline"] = base
            # free parity check: the GPU path on the very same prefix must give the oracle's table
            wl.step(0, min(m, args.reads))
            g_counts, g_total, g_matched = wl.result()
            if m <= args.reads:
                ok = (g_counts.tolist() == ctr.table() and g_matched == ctr.matched_reads() and g_total == ctr.total_reads())
                out["parity_vs_oracle_on_cpu_sample"] = "bit-exact" if ok else "MISMATCH"
            out["speedup_vs_cpu_baseline"] = out["value"] / base["value"]
        print(json.dumps(out), flush=True)
    wl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
