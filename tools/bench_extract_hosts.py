"""Host side of the sharded extraction at N ranks on ONE host, without GPUs: does the host keep up with 8 GPUs?

Every rank runs esm_amd.extract.extract() exactly as on the MI355X — FASTA strings -> LUT tokeniser -> (stub) forward
-> per-sequence slices -> writer threads -> one .pt file per sequence in the reference's format
(scripts/extract.py:104-131) — with the forward replaced by a stub that hands back a [B, T, E] fp32 tensor after
`--gpu-ms` milliseconds (the time one MI355X needs for the batch: 64 x 1022 tokens ~ 100 ms).  The ranks are gloo
processes pinned to their share of the CPUs by esm_amd.launch.pin_rank_cpus, as the real launch does.

    python tools/bench_extract_hosts.py --world 8 [--seqs-per-rank 256] [--len 1022] [--embed-dim 1280] [--gpu-ms 100]

Prints one JSON line: files/s and residues/s per rank and in total, and what 8 GPUs at 650 k residues/s each would need.
"""
import argparse
import json
import os
import pathlib
import shutil
import sys
import tempfile
import time

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _worker(rank, world, port, args, out_dir, q):
    import torch.distributed as dist

    from esm_amd import Alphabet, FastaBatchedDataset
    from esm_amd.extract import extract
    from esm_amd.launch import pin_rank_cpus

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if getattr(args, "no_crc32", False) and hasattr(torch.serialization, "set_crc32_options"):
        torch.serialization.set_crc32_options(False)
    cpus = pin_rank_cpus(rank, world, force=True) if not args.no_affinity else set()
    torch.set_num_threads(max(1, min(8, len(cpus) or 8)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        aas = "LAGVSERTIDPKQNFYMHWC"
        n = args.seqs_per_rank * world
        seqs = ["".join(aas[i] for i in torch.randint(0, 20, (args.len,), generator=g).tolist()) for _ in range(16)]
        ds = FastaBatchedDataset([f"s{i}" for i in range(n)], [seqs[i % 16] for i in range(n)])
        alphabet = Alphabet.from_architecture("ESM-1b")
        E = args.embed_dim
        base = torch.randn(args.len + 2, E, generator=torch.Generator().manual_seed(11))

        def stub(toks, layers, return_contacts):
            t0 = time.perf_counter()
            reps = {l: (base[: toks.shape[1]].unsqueeze(0) + toks[:, :, None].float()).contiguous() for l in layers}
            left = args.gpu_ms / 1e3 - (time.perf_counter() - t0)
            if left > 0:
                time.sleep(left)  # the GPU's time for the batch; the host thread is free meanwhile
            return {"representations": reps}

        dist.barrier()
        t0 = time.perf_counter()
        means = extract(ds, alphabet, stub, num_layers=33, embed_dim=E, repr_layers=[33], include=args.include,
                        output_dir=pathlib.Path(out_dir), toks_per_batch=args.toks_per_batch, log=lambda s: None,
                        async_host=True, writer_threads=args.writer_threads, gather_mean=True)
        dt = time.perf_counter() - t0
        dist.barrier()
        total = time.perf_counter() - t0
        if args.check:
            ids = torch.tensor(alphabet.encode(seqs[3]), dtype=torch.float32)
            want = (base[1:args.len + 1] + ids[:, None]).mean(0)
            assert torch.allclose(means[33][3], want, atol=1e-4), "gathered mean embedding differs"
        q.put((rank, dt, total, args.seqs_per_rank, len(cpus)))
    finally:
        dist.destroy_process_group()


def run(args):
    out_dir = tempfile.mkdtemp(prefix="esm_hosts_", dir=args.out_root)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        from esm_amd.launch import free_port

        port = free_port()
        procs = [ctx.Process(target=_worker, args=(r, args.world, port, args, out_dir, q)) for r in range(args.world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        rows = sorted(q.get() for _ in range(args.world))
        n_files = len(list(pathlib.Path(out_dir).glob("*.pt")))
        assert n_files == args.seqs_per_rank * args.world, (n_files, args.seqs_per_rank * args.world)
        wall = max(r[2] for r in rows)
        res = {
            "world": args.world, "host_cpus": os.cpu_count(), "cpus_per_rank": rows[0][4], "files": n_files,
            "files_per_s_total": round(n_files / wall, 1),
            "files_per_s_per_rank": [round(r[3] / r[1], 1) for r in rows],
            "residues_per_s_total": round(n_files * args.len / wall, 1),
            "file_MB": round(args.len * args.embed_dim * 4 / 1e6, 2) if "per_tok" in args.include else 0.0,
            "gpu_ms_per_batch": args.gpu_ms, "include": args.include, "crc32": not bool(getattr(args, "no_crc32", False)), "writer_threads": args.writer_threads,
            "gpu_bound_residues_per_s_total": round(args.world * args.toks_per_batch / (args.gpu_ms / 1e3), 1) if args.gpu_ms else None,
        }
        return res
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--seqs-per-rank", type=int, default=256)
    ap.add_argument("--len", type=int, default=1022)
    ap.add_argument("--embed-dim", type=int, default=1280)
    ap.add_argument("--toks-per-batch", type=int, default=65536)
    ap.add_argument("--gpu-ms", type=float, default=100.0)
    ap.add_argument("--writer-threads", type=int, default=0)
    ap.add_argument("--include", nargs="+", default=["mean", "per_tok"])
    ap.add_argument("--out-root", default="/dev/shm" if os.path.isdir("/dev/shm") else None)
    ap.add_argument("--no-affinity", action="store_true")
    ap.add_argument("--no_crc32", action="store_true", help="torch.save without the zip CRC32 (the driver's --no_crc32; its default keeps the checksum)")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    print(json.dumps(run(args)), flush=True)


if __name__ == "__main__":
    main()
