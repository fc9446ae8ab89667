"""Sharded bulk extraction (SURVEY.md §8 e): the batch plan, the LPT assignment and the RCCL/gloo
gather are host logic and run on CPU; the model forward is replaced by a deterministic stub here (the
real forward is covered on the MI355X by tests/test_model_gpu.py)."""
import os
import pathlib
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from esm_amd import Alphabet, FastaBatchedDataset  # noqa: E402
from esm_amd.extract import assign_batches, batch_cost, extract  # noqa: E402

E = 16
LENGTHS = [428, 222, 502, 156, 80, 71, 99, 602, 127, 152, 98, 935, 102, 882, 659]  # some_proteins.fasta


def _dataset():
    g = torch.Generator().manual_seed(5)
    aas = "LAGVSERTIDPKQNFYMHWC"
    seqs = ["".join(aas[i] for i in torch.randint(0, 20, (n,), generator=g).tolist()) for n in LENGTHS]
    return FastaBatchedDataset([f"seq{i}" for i in range(len(seqs))], seqs)


def _stub_embed(toks, layers, return_contacts):
    """representation[l][b,t,:] = token id * (l+1) + channel index: depends only on the sequence."""
    ch = torch.arange(E, dtype=torch.float32)
    reps = {l: toks[:, :, None].float() * (l + 1) + ch for l in layers}
    out = {"representations": reps, "logits": torch.zeros(toks.shape + (33,))}
    if return_contacts:
        T = toks.shape[1]
        out["contacts"] = torch.zeros((toks.shape[0], T - 2, T - 2))
    return out


def _expected_means(ds, alphabet, layer):
    rows = []
    for _, s in ds:
        ids = torch.tensor(alphabet.encode(s), dtype=torch.float32)
        rows.append((ids[:, None] * (layer + 1) + torch.arange(E, dtype=torch.float32)).mean(0))
    return torch.stack(rows)


def test_assign_batches_covers_everything_and_balances():
    ds = _dataset()
    batches = ds.get_batch_indices(1024, extra_toks_per_seq=1)
    # golden: the reference's batching of these lengths (SURVEY.md §8 c)
    assert batches == [[5, 4, 10, 6, 12, 8], [9, 3, 1], [0, 2], [7], [14], [13], [11]]
    for world in (1, 2, 3, 8):
        plan = assign_batches(batches, LENGTHS, world, 1280)
        flat = sorted(b for r in plan for b in r)
        assert flat == list(range(len(batches)))  # every batch exactly once
        assert plan == assign_batches(batches, LENGTHS, world, 1280)  # deterministic
        cost = lambda bid: batch_cost(len(batches[bid]), max(LENGTHS[i] for i in batches[bid]) + 2, 1280)
        loads = [sum(cost(b) for b in r) for r in plan]
        biggest = max(cost(b) for b in range(len(batches)))
        assert max(loads) - min(loads) <= biggest + 1e-6  # LPT bound


def test_single_process_extract_matches_reference_file_format(tmp_path):
    ds = _dataset()
    alphabet = Alphabet.from_architecture("ESM-1b")
    means = extract(ds, alphabet, _stub_embed, num_layers=6, embed_dim=E, repr_layers=[-1, 0], include=["mean", "per_tok", "bos"],
                    output_dir=tmp_path, toks_per_batch=1024, log=lambda s: None)
    assert sorted(means) == [0, 6]
    assert torch.allclose(means[6], _expected_means(ds, alphabet, 6))
    r = torch.load(tmp_path / "seq3.pt")
    assert r["label"] == "seq3"
    assert r["representations"][6].shape == (LENGTHS[3], E)
    assert torch.allclose(r["mean_representations"][0], means[0][3])
    assert r["bos_representations"][6].shape == (E,)


def _worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = _dataset()
        alphabet = Alphabet.from_architecture("ESM-1b")
        means = extract(ds, alphabet, _stub_embed, num_layers=6, embed_dim=E, repr_layers=[6], include=["mean"],
                        output_dir=pathlib.Path(tmp), toks_per_batch=1024, log=lambda s: None)
        exp = _expected_means(ds, alphabet, 6)
        assert torch.allclose(means[6], exp), f"rank {rank}: gathered means differ"
        dist.barrier()
        if rank == 0:
            files = sorted(p.name for p in pathlib.Path(tmp).glob("*.pt"))
            assert files == sorted(f"seq{i}.pt" for i in range(len(ds)))  # every sequence written exactly once
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_extract_gloo(tmp_path, world):
    from esm_amd.launch import free_port

    port = free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)


def test_host_side_at_eight_ranks(tmp_path):
    """8 ranks x (tokeniser + chunked writer threads) on ONE host, the real extract() pipeline
    with the writer threads active (async_host) and a stub forward; every file written exactly once, the gathered
    means right on every rank.  tools/bench_extract_hosts.py is the same harness at full size (5 MB files,
    64 k-token batches, 100 ms per batch) for the GPU host's 256 threads; its number is quoted in DESIGN.md §6."""
    import argparse
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_extract_hosts as bh

    args = argparse.Namespace(world=8, seqs_per_rank=12, len=150, embed_dim=32, toks_per_batch=1024, gpu_ms=2.0,
                              writer_threads=3, include=["mean", "per_tok"], out_root=str(tmp_path), no_affinity=False,
                              check=True)
    res = bh.run(args)
    assert res["files"] == 96 and len(res["files_per_s_per_rank"]) == 8
    assert res["files_per_s_total"] > 0


def test_rank_cpu_slices_partition_the_host():
    from esm_amd.launch import rank_cpu_slice

    for n, world in ((256, 8), (64, 4), (8, 8), (6, 4)):
        slices = [rank_cpu_slice(r, world, range(n)) for r in range(world)]
        if n >= 2 * world:
            assert all(len(s) == n // world for s in slices)
            assert set().union(*slices) == set(range(n)) and sum(len(s) for s in slices) == n  # disjoint, complete
        else:
            assert all(s == set(range(n)) for s in slices)  # too few CPUs to split: everyone keeps all


def test_pin_rank_cpus_slices_a_shared_set_and_keeps_a_per_rank_set(monkeypatch):
    """a cpuset shared by all ranks of a job (larger than one rank's share of the host) is sliced per rank; a
    set that already is at most one rank's share is kept; ESM_AMD_NO_AFFINITY switches everything off."""
    import os

    from esm_amd import launch

    state = {"aff": set(range(64))}
    monkeypatch.setattr(os, "cpu_count", lambda: 256)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(state["aff"]))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: state.__setitem__("aff", set(cpus)))
    monkeypatch.delenv("ESM_AMD_NO_AFFINITY", raising=False)
    # 64 of 256 CPUs for a job of 8 ranks: shared (> 256 / 8) -> rank 3 gets its eighth of the 64
    got = launch.pin_rank_cpus(3, 8)
    assert len(got) == 8 and got <= set(range(64)) and got == launch.rank_cpu_slice(3, 8, range(64))
    # 32 CPUs = exactly one rank's share of the host: a per-rank set, kept
    state["aff"] = set(range(32, 64))
    assert launch.pin_rank_cpus(3, 8) == set(range(32, 64))
    # ... unless forced (the host benchmark starts its ranks from one unrestricted parent)
    assert len(launch.pin_rank_cpus(3, 8, force=True)) == 4
    # the whole host
    state["aff"] = set(range(256))
    assert len(launch.pin_rank_cpus(0, 8)) == 32
    # with the peers' sets (init_ranks gathers them): torchrun NUMA binding, 2 nodes x 4 ranks — ranks 4..7 share the second
    # node's 128 hardware threads; rank 5 gets the second quarter OF THAT SET (32 CPUs), not an eighth of it
    node = [set(range(0, 64)) | set(range(128, 192)), set(range(64, 128)) | set(range(192, 256))]
    peers = [sorted(node[r // 4]) for r in range(8)]
    state["aff"] = set(node[1])
    got = launch.pin_rank_cpus(5, 8, peers=peers)
    assert len(got) == 32 and got <= node[1] and got == launch.rank_cpu_slice(1, 4, node[1])
    # per-rank sets (all different): kept; one cpuset shared by all 8: an eighth each
    state["aff"] = set(range(40, 80))
    assert launch.pin_rank_cpus(1, 8, peers=[sorted(range(40 * r, 40 * r + 40)) for r in range(8)]) == set(range(40, 80))
    state["aff"] = set(range(64))
    assert launch.pin_rank_cpus(3, 8, peers=[sorted(range(64))] * 8) == launch.rank_cpu_slice(3, 8, range(64))
    state["aff"] = set(range(256))
    monkeypatch.setenv("ESM_AMD_NO_AFFINITY", "1")
    assert launch.pin_rank_cpus(0, 8) == set(range(256))


def test_writer_runs_a_batch_on_all_threads():
    """a batch is several chunk jobs, so more than `depth` threads can work at once and `done` runs once."""
    import threading
    import time

    from esm_amd.extract import _Writer

    w = _Writer(depth=1, threads=4)
    seen, done = set(), []
    lock = threading.Lock()

    def job():
        with lock:
            seen.add(threading.get_ident())
        time.sleep(0.05)

    w.submit([job] * 8, done=lambda: done.append(1))
    w.submit([job] * 8, done=lambda: done.append(2))  # waits for the first batch's slot
    w.close()
    assert len(seen) == 4 and done == [1, 2]
