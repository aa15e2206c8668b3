"""GPU: the N > 1 flows of BASELINE.json configs[4] (one sample per GPU, one exchange of the count matrix).

* `python bench.py --gpus 2` invoked directly (self-launching: fresh children before any GPU call), rehearsed over
  gloo so that it also runs on a one-GPU box (both ranks share the card);
* the same over RCCL ("nccl"), and the C++ command line dealing two samples to two devices — both need two GPUs and
  are skipped otherwise.
The reference's counterpart is the rayon pool over samples, src/count.rs:117-136."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


SMALL = ["--reads", "3000000", "--guides", "20000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--e2e-reads", "0"]


def test_bench_self_launch_two_ranks_gloo_rehearsal():
    one = _bench({}, "--gpus", "1", *SMALL)
    two = _bench({"SGC_BENCH_BACKEND": "gloo"}, "--gpus", "2", *SMALL)
    assert one["n_gpus"] == 1 and one["launch"] == "single process" and "exchange" not in one
    assert two["n_gpus"] == 2 and two["launch"] == "self-launched children"
    ex = two["exchange"]
    assert ex["backend"] == "gloo" and ex["ranks_seen"] == 2
    assert sorted(d["rank"] for d in ex["rank_devices"]) == [0, 1]
    assert ex["all_gather_batch_ms"] > 0 and ex["all_gather_one_sample_ms"] > 0
    assert ex["batch_bytes_per_rank"] == 3 * (20000 + 2) * 8
    # same per-rank workload: rank 0's sample is the N = 1 sample, so the matched fraction agrees exactly
    assert two["matched_fraction"] == one["matched_fraction"]
    assert two["config"]["reads_per_gpu"] == 3000000 and two["scaling"] == "weak"
    # N > 1: rank 0 also runs N samples through ONE command line dealt to the N devices (here: both to the one card), the host-bound curve
    ms = two["e2e"]["multi_sample"]
    assert ms["samples"] == 2 and ms["worker_threads"] == 2 and ms["reads_counted"] == [1500000, 1500000], ms
    assert ms["reads_per_s"] > 0 and len(ms["sample_s"]) == 2


RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from sgcount_amd.workload import DeviceWorkload
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), world_size=1, rank=0,
                        device_id=torch.device("cuda", 0))
wl = DeviceWorkload(200000, 2000, 20, one_mismatch=True)
row = torch.zeros(2002, dtype=torch.int64, device=wl.dev)
wl.step(out=row)                                                  # our kernels and the collectives on one stream
gathered = torch.empty((1, 2002), dtype=torch.int64, device=wl.dev)
dist.all_gather_into_tensor(gathered.view(-1), row)               # the exchange of bench.py --gpus N, over RCCL
summed = row.clone()
dist.all_reduce(summed, op=dist.ReduceOp.SUM)                     # the within-sample reduce of sgcount_amd.distributed
dist.barrier()
torch.cuda.synchronize()
counts, total, matched = wl.result(row)
assert total == 200000 and 0 < matched <= total and int(counts.sum()) == matched
assert torch.equal(gathered[0], row) and torch.equal(summed, row)
dist.destroy_process_group()
print("rccl ok", total, matched)
"""


def test_rccl_collectives_on_one_rank():
    """What a one-GPU box can show of the RCCL path: the process group comes up on the device, and the two collectives the N > 1 flows use
    (all-gather of the count rows, sum of partial rows) run on the stream our kernels ran on and leave the row intact."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", RCCL_ONE_RANK % ROOT, str(port)], capture_output=True, timeout=300, env=env)
    assert p.returncode == 0 and b"rccl ok 200000" in p.stdout, (p.stdout.decode()[-500:], p.stderr.decode()[-2000:])


def test_bench_two_ranks_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    two = _bench({}, "--gpus", "2", *SMALL)
    ex = two["exchange"]
    assert ex["backend"] == "nccl" and ex["ranks_seen"] == 2
    assert sorted(d["device"] for d in ex["rank_devices"]) == [0, 1]


def test_cli_two_samples_over_two_devices(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from sgcount_amd import hostlib, synth
    lib = synth.library(2000, 20)
    lib_text = synth.library_fasta(lib)
    lp = str(tmp_path / "lib.fa")
    open(lp, "wb").write(lib_text)
    paths, texts = [], []
    for i in range(4):
        t = synth.fastq_host(lib, 0, 30000 + 1000 * i, seed=synth.READS_SEED + i)
        p = str(tmp_path / ("s%d.fastq" % i))
        open(p, "wb").write(t)
        paths.append(p); texts.append(t)
    out, stats = str(tmp_path / "out.tsv"), str(tmp_path / "stats.json")
    p = subprocess.run([hostlib.cli_path(), "-l", lp, "-i", *paths, "-a", "30", "-q", "-t", "2", "-o", out, "--stats-json", stats],
                       capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    st = json.load(open(stats))
    assert st["devices"] >= 2 and {s["device"] for s in st["samples"]} == {0, 1}
    rows = [ln.split("\t") for ln in open(out).read().splitlines()[1:]]
    got = {r[0]: [int(x) for x in r[1:]] for r in rows}
    for i, t in enumerate(texts):
        want, _, _ = O.count_text(lib_text, t, False, 30, False, True)
        for g, c in enumerate(want):
            assert got.get("sg%06d" % g, [0] * 4)[i] == c


def test_cli_workers_on_one_device_share_one_table_set(tmp_path):
    """`-t 4` on one GPU: four contexts, ONE table build (the others are clones that share the device's tables — the reference
    builds Library and Permuter once and lends them to every rayon worker, count.rs:103-136).  The set-up of four contexts
    must not cost four table builds: less than 1.5x the set-up of one (it was 4x), and the table is the oracle's."""
    from sgcount_amd import hostlib, synth
    lib = synth.library(100_000, 20)
    lib_text = synth.library_fasta(lib)
    lp = str(tmp_path / "lib.fa")
    open(lp, "wb").write(lib_text)
    paths, texts = [], []
    for i in range(4):
        t = synth.fastq_host(lib, 0, 20000 + 500 * i, seed=synth.READS_SEED + i)
        p = str(tmp_path / ("s%d.fastq" % i))
        open(p, "wb").write(t)
        paths.append(p); texts.append(t)
    out, stats = str(tmp_path / "out.tsv"), str(tmp_path / "stats.json")

    def run(threads):
        p = subprocess.run([hostlib.cli_path(), "-l", lp, "-i", *paths, "-a", "30", "-q", "-t", str(threads), "--devices", "1", "-o", out,
                            "--stats-json", stats], capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        return json.load(open(stats))
    run(1)                                             # warm the page cache / the device
    one = min((run(1) for _ in range(2)), key=lambda s: s["contexts_ready_s"])
    four = min((run(4) for _ in range(2)), key=lambda s: s["contexts_ready_s"])
    assert one["contexts"] == 1 and four["contexts"] == 4 and four["devices"] == 1
    assert len(four["table_build_per_device_s"]) == 1
    # (contexts_ready_s: device init + table build + the clones of the other worker threads; table_build_s alone is one device's build)
    assert four["contexts_ready_s"] < 1.5 * one["contexts_ready_s"] + 0.05, (one["contexts_ready_s"], four["contexts_ready_s"])
    rows = [ln.split("\t") for ln in open(out).read().splitlines()[1:]]
    got = {r[0]: [int(x) for x in r[1:]] for r in rows}
    want0, _, _ = O.count_text(lib_text, texts[0], False, 30, False, True)
    for g, c in enumerate(want0):
        assert got.get("sg%06d" % g, [0] * 4)[0] == c
