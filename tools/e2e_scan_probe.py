#!/usr/bin/env python3
"""Sweeps sgcount-hip options on the bench's 100M-read FASTQ text (page cache -> table): one line per run.
python tools/e2e_scan_probe.py [reads] -- "<extra args of run 1>" "<extra args of run 2>" ..."""
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402


def main():
    argv = sys.argv[1:]
    n = 100_000_000
    if argv and argv[0] != "--":
        n = int(argv.pop(0))
    configs = argv[1:] if argv and argv[0] == "--" else [""]
    from sgcount_amd import hostlib, synth
    from sgcount_amd.workload import DeviceWorkload
    wl = DeviceWorkload(1_000_000, 100_000, 20, one_mismatch=True, position_recursion=True, offset=30, reads_seed=synth.READS_SEED)
    d = "/dev/shm/sgc_probe_%d" % os.getpid()
    os.makedirs(d, exist_ok=True)
    try:
        lib_path, fq, table = os.path.join(d, "library.fa"), os.path.join(d, "reads.fastq"), os.path.join(d, "table.tsv")
        open(lib_path, "wb").write(synth.library_fasta(wl.lib_seqs))
        t0 = time.perf_counter()
        bench.write_fastq(wl, n, fq)
        print("wrote %.1f GB in %.1f s" % (os.path.getsize(fq) / 1e9, time.perf_counter() - t0), flush=True)
        cli = hostlib.cli_path()
        base = ["-l", lib_path, "-a", "30", "-q", "-o", table, "-i", fq]
        ref = None
        for cfg in configs:
            for rep in range(2):
                wall, st = bench._run_cli(cli, base + cfg.split(), stats=os.path.join(d, "stats.json"))
                tab = open(table, "rb").read()
                ref = ref or tab
                s0 = st["samples"][0]
                print("%-40s wall %.3f  start %.3f lib %.3f init %.3f tables %.3f sample %.3f (wait_text %.3f busy/thr %.3f thr %d copy %.3f) free %.3f exit %.3f  %s" % (
                    cfg or "(default)", wall, st["process_start_to_count_s"], st["library_load_s"], st["device_init_s"],
                    st["table_build_s"], s0["wall_s"], s0["wait_for_text_s"], s0["read_busy_s"] / max(s0["reader_threads"], 1),
                    s0["reader_threads"], s0["host_copy_s"], st["context_free_s"], st["teardown_s"], "same table" if tab == ref else "TABLE DIFFERS"), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
        wl.close()


main()
