"""world_size-2 gloo test of the multi-GPU orchestration (sample sharding + count-matrix exchange).
The per-sample counts come from the CPU oracle here (no GPU in this container); on the GPU box the same
helper carries device tensors over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sample_rows(n_samples):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import _oracle as O
    from sgcount_amd import synth
    lib = synth.library(300, 20)
    lib_text = synth.library_fasta(lib)
    rows = {}
    for i in range(n_samples):
        fq = synth.fastq_host(lib, 0, 2000 + 100 * i, seed=synth.READS_SEED + i)
        counts, tot, mat = O.count_text(lib_text, fq, False, 30, False, True)
        rows[i] = counts + [tot, mat]
    return rows


def _worker(rank, world, port, n_samples, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(HERE))
    from sgcount_amd.distributed import assign_samples, gather_count_matrix
    all_rows = _sample_rows(n_samples)
    mine = assign_samples(n_samples, world, rank)
    local = {i: torch.tensor(all_rows[i], dtype=torch.int64) for i in mine}
    mat = gather_count_matrix(local, n_samples, 302)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), mat.numpy())
    if n_samples == world:          # one sample per rank: the single all-gather path bench.py uses
        from sgcount_amd.distributed import all_gather_rows
        assert torch.equal(all_gather_rows(local[rank]), mat)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_samples", [5, 2, 1])
def test_two_rank_count_matrix(tmp_path, n_samples):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_samples, str(tmp_path)), nprocs=world, join=True)
    want = np.array([_sample_rows(n_samples)[i] for i in range(n_samples)], dtype=np.int64)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want)


def test_assign_samples_round_robin():
    from sgcount_amd.distributed import assign_samples
    assert assign_samples(5, 2, 0) == [0, 2, 4] and assign_samples(5, 2, 1) == [1, 3]
    assert assign_samples(1, 8, 3) == [] and assign_samples(8, 8, 7) == [7]
    seen = sorted(i for r in range(4) for i in assign_samples(10, 4, r))
    assert seen == list(range(10))


def _worker_within(rank, world, port, n_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import _oracle as O
    from sgcount_amd import synth
    from sgcount_amd.distributed import reduce_sample_row, shard_reads
    lib = synth.library(300, 20)
    first, count = shard_reads(n_reads, world, rank)
    counts, tot, mat = O.count_text(synth.library_fasta(lib), synth.fastq_host(lib, first, count), False, 30, False, True) \
        if count else ([0] * 300, 0, 0)
    row = reduce_sample_row(torch.tensor(counts + [tot, mat], dtype=torch.int64))
    np.save(os.path.join(out_dir, "within%d.npy" % rank), row.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_reads", [3001, 1])
def test_two_rank_within_sample(tmp_path, n_reads):
    """One sample, reads dealt to two ranks, partial rows summed: equals counting the whole sample at once."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import _oracle as O
    from sgcount_amd import synth
    world, port = 2, _free_port()
    mp.spawn(_worker_within, args=(world, port, n_reads, str(tmp_path)), nprocs=world, join=True)
    lib = synth.library(300, 20)
    counts, tot, mat = O.count_text(synth.library_fasta(lib), synth.fastq_host(lib, 0, n_reads), False, 30, False, True)
    for r in range(world):
        assert np.load(os.path.join(str(tmp_path), "within%d.npy" % r)).tolist() == counts + [tot, mat]


def test_shard_reads_covers_everything():
    from sgcount_amd.distributed import shard_reads
    for n, w in ((10, 4), (3, 8), (0, 2), (100_000_000, 8)):
        parts = [shard_reads(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
