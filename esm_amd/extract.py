"""Bulk FASTA embedding extraction, sharded over the GPUs of one node.

Mirrors the reference driver ``scripts/extract.py`` (arguments, batching, the per-sequence ``.pt`` result
files: reference scripts/extract.py:15-131) and adds the data-parallel split of SURVEY.md §8 e:

* sequences are independent units, so there is no exchange step in the math: weights are replicated, the
  length-sorted token-budget batches of ``FastaBatchedDataset.get_batch_indices`` (reference
  esm/data.py:65-88) are assigned to ranks by longest-processing-time on the algorithmic cost
  ``B*T*(24 E^2 + 4 T E)`` per layer, each rank writes the result files of its own sequences;
* the only collectives (RCCL over xGMI on the GPUs, gloo in the CPU tests) are the scatter/gather the
  north star names: an all-gather of the fixed-size mean embeddings so that rank 0 can return / write
  one ``[n_sequences, E]`` matrix per layer in FASTA order.

    python -m esm_amd.extract model.pt seqs.fasta out/ --repr_layers 33 --include mean per_tok
    python -m esm_amd.extract model.pt seqs.fasta out/ --repr_layers 33 --include mean --gpus 8   # spawns 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m esm_amd.extract model.pt seqs.fasta out/ --repr_layers 33 --include mean

The model forward is ``esm_amd.ESM2`` on the MI355X (no CPU fallback); ``embed_fn`` exists so that the
sharding / gather logic can be exercised by the world_size-2 gloo tests without a GPU.
"""
import argparse
import os
import pathlib
import queue
import sys
import threading
from typing import Callable, Dict, List, Optional, Sequence

import torch


def batch_cost(n_seqs: int, width: int, embed_dim: int) -> float:
    """Algorithmic FLOPs of one layer on a padded batch (SURVEY.md §8 a: 24 E^2 + 4 T E per token)."""
    return float(n_seqs) * width * (24.0 * embed_dim * embed_dim + 4.0 * width * embed_dim)


def assign_batches(batches: Sequence[Sequence[int]], lengths: Sequence[int], world: int, embed_dim: int,
                   extra_toks: int = 2) -> List[List[int]]:
    """Longest-processing-time assignment of batches (lists of sequence indices) to ``world`` ranks.
    Returns, per rank, the list of batch ids in the order they should run (largest first).
    Deterministic: ties are broken by batch id and rank id, so every rank computes the same plan."""
    costs = []
    for bid, b in enumerate(batches):
        width = max(lengths[i] for i in b) + extra_toks
        costs.append((batch_cost(len(b), width, embed_dim), bid))
    order = sorted(costs, key=lambda c: (-c[0], c[1]))
    load = [0.0] * world
    plan: List[List[int]] = [[] for _ in range(world)]
    for cost, bid in order:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(bid)
        load[r] += cost
    return plan


def _dist_info():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def gather_rows(local_rows: torch.Tensor, local_index: torch.Tensor, n_total: int) -> Optional[torch.Tensor]:
    """All-gather of fixed-width rows: every rank contributes ``local_rows [n_i, W]`` for the global
    row ids ``local_index [n_i]``; returns the ``[n_total, W]`` matrix (on every rank).  Counts differ
    between ranks, so rows are padded to the maximum count before the (equal-size) all_gather."""
    dist, rank, world = _dist_info()
    W = local_rows.shape[1]
    if world == 1:
        out = local_rows.new_zeros((n_total, W))
        out[local_index] = local_rows
        return out
    dev = local_rows.device
    n_local = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    n_max = int(max(c.item() for c in counts))
    pad_rows = local_rows.new_zeros((n_max, W))
    pad_rows[: local_rows.shape[0]] = local_rows
    pad_idx = torch.full((n_max,), -1, dtype=torch.int64, device=dev)
    pad_idx[: local_index.shape[0]] = local_index.to(dev)
    all_rows = [torch.empty_like(pad_rows) for _ in range(world)]
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    dist.all_gather(all_rows, pad_rows)
    dist.all_gather(all_idx, pad_idx)
    out = local_rows.new_zeros((n_total, W))
    for rows, idx, c in zip(all_rows, all_idx, counts):
        n = int(c.item())
        out[idx[:n]] = rows[:n]
    return out


def default_writer_threads() -> int:
    """``torch.save`` of a 5 MB per-sequence result costs ~4 ms of host time (pickle + CRC32 + copy into the zip
    container), most of it with the GIL released, so writer THREADS scale — up to a point: 277 / 558 / 1063 / 993 /
    912 files/s with 1 / 2 / 4 / 8 / 16 threads on an idle 8-core host, and 24 threads on the 256-thread GPU host
    were slower than 8 (the pickling parts hold the GIL: more threads, longer convoys).  440 files/s keep one
    MI355X busy at L = 1022.  With 8 ranks on one host the limit is the kernel's page-cache write path, not this
    process (~7 GB/s of new file pages per host whatever the file system; profiles/r4_extract_hosts.log): 4 threads per
    rank wrote 1.44 M residues/s there, 8 threads 1.31 M, 16 threads 1.25 M — so several ranks on one host take 4."""
    n = max(2, min(8, (os.cpu_count() or 4) // 2))
    try:
        if int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) >= 4:
            n = min(n, 4)
    except ValueError:
        pass
    return n


class _Writer:
    """Background result writer: the device->host copy of batch i (on its own HIP stream, into pinned
    buffers) and the slicing / ``torch.save`` of its sequences overlap the forward pass of batch i+1
    (SURVEY.md §8 d: 335 MB of fp32 per 64 x 1022 batch would otherwise serialise behind the compute).
    A batch is submitted as several CHUNK jobs (a few sequences each), so that all ``threads`` work on it at once
    (``torch.save`` releases the GIL for most of its time); at most ``depth`` batches are in flight, which bounds
    the pinned memory; ``done`` runs once after a batch's last chunk."""

    def __init__(self, depth=3, threads=2):
        self.q = queue.Queue()
        self.slots = threading.Semaphore(depth)
        self.errors: List[BaseException] = []
        self.lock = threading.Lock()
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(threads)]
        for t in self.threads:
            t.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            job, state = item
            try:
                job()
            except BaseException as e:  # surfaced by close()
                self.errors.append(e)
            finally:
                with self.lock:
                    state["left"] -= 1
                    last = state["left"] == 0
                if last:
                    try:
                        if state["done"] is not None:
                            state["done"]()
                    except BaseException as e:
                        self.errors.append(e)
                    finally:
                        self.slots.release()

    def submit(self, jobs, done=None):
        """``jobs``: the chunk jobs of ONE batch (a single callable is one chunk)."""
        jobs = [jobs] if callable(jobs) else list(jobs)
        self.slots.acquire()
        if self.errors or not jobs:
            self.slots.release()
            if self.errors:
                raise self.errors[0]
            return
        state = {"left": len(jobs), "done": done}
        for job in jobs:
            self.q.put((job, state))

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.errors:
            raise self.errors[0]


class _PinnedPool:
    """Reusable pinned host buffers keyed by (shape, dtype): hipHostMalloc per batch would cost more than the copy."""

    def __init__(self):
        self.free: Dict[tuple, List[torch.Tensor]] = {}
        self.lock = threading.Lock()

    def take(self, like: torch.Tensor) -> torch.Tensor:
        key = (tuple(like.shape), like.dtype)
        with self.lock:
            lst = self.free.get(key)
            if lst:
                return lst.pop()
        return torch.empty(like.shape, dtype=like.dtype, pin_memory=True)

    def give(self, t: torch.Tensor):
        with self.lock:
            self.free.setdefault((tuple(t.shape), t.dtype), []).append(t)


def extract(dataset, alphabet, embed_fn: Callable[[torch.Tensor, List[int], bool], Dict], num_layers: int,
            embed_dim: int, repr_layers: Sequence[int], include: Sequence[str],
            output_dir: Optional[pathlib.Path] = None, toks_per_batch: int = 4096,
            truncation_seq_length: int = 1022, device: Optional[torch.device] = None,
            gather_mean: bool = True, log: Callable[[str], None] = print, writer_threads: int = 0,
            writer_depth: int = 4, async_host: bool = False, chunk_rows: int = 8):
    """Run the sharded extraction.  ``embed_fn(tokens, repr_layers, return_contacts)`` is the model
    forward (``ESM2.__call__`` / ``ESM2.forward_varlen`` in production; an ``embed_fn.wants_lengths = True``
    attribute asks for the extra keyword ``lengths`` = per-row token counts, taken from the host copy).  Returns ``{layer: [n_sequences, E] mean embeddings}``
    in dataset order when ``gather_mean`` (every rank gets the full matrix), else ``{}``.
    ``async_host``: run the writer-thread pipeline also without a GPU (host-side scaling tests: 8 ranks x (tokeniser +
    writers) on one host, tests/test_extract_sharded.py); ``chunk_rows``: sequences per writer job."""
    dist, rank, world = _dist_info()
    assert all(-(num_layers + 1) <= i <= num_layers for i in repr_layers)
    # duplicates (e.g. --repr_layers -1 33) collapse, first occurrence wins: the reference builds dicts keyed by layer
    layers = list(dict.fromkeys((i + num_layers + 1) % (num_layers + 1) for i in repr_layers))
    return_contacts = "contacts" in include
    lengths = [min(len(s), truncation_seq_length) for s in dataset.sequence_strs]
    batches = dataset.get_batch_indices(toks_per_batch, extra_toks_per_seq=1)
    plan = assign_batches(batches, lengths, world, embed_dim)
    convert = alphabet.get_batch_converter(truncation_seq_length)
    if output_dir is not None:
        output_dir.mkdir(parents=True, exist_ok=True)

    my_index: List[int] = []
    my_means: Dict[int, List[torch.Tensor]] = {l: [] for l in layers}
    lock = threading.Lock()

    def batch_means(reps, strs):
        """`t[i, 1 : n+1].mean(0)` of every sequence (reference scripts/extract.py:113-116) for the whole batch on the
        tensors' device: one pass of ``esmk_op_masked_row_mean`` per layer (csrc/elementwise.hip) on the GPU; a
        per-sequence ``mean(0)`` on the host costs ~20 ms per 1022 x 1280 slice.  The slice is cut at the tensor's
        end and an empty one gives NaN, as in the reference; the result has the dtype of the representations."""
        any_t = next(iter(reps.values()))
        n = torch.tensor([min(truncation_seq_length, len(s)) for s in strs], dtype=torch.int32)
        if any_t.is_cuda:
            from . import ops

            n = n.to(any_t.device, non_blocking=True)
            return {l: ops.masked_row_mean(t.contiguous(), n, first_row=1).to(t.dtype) for l, t in reps.items()}
        # CPU tensors only reach this function in the gloo sharding tests (stub embed_fn), never on the product path
        return {l: torch.stack([t[i, 1:int(k) + 1].float().mean(0) for i, k in enumerate(n)]).to(t.dtype)
                for l, t in reps.items()}

    def finish(ids, labels, strs, reps, contacts, means_b, rows=None):
        """Per-sequence results of (rows ``rows`` of) one batch from HOST tensors (reference scripts/extract.py:104-131)."""
        rows_idx, rows_mean = [], {l: [] for l in layers}
        # torch.save of a VIEW writes the view's whole storage (the batch).  A slice [row, a:b] of the contiguous
        # host tensor is itself one contiguous run of memory: re-wrapping it (numpy view -> from_numpy; bf16 goes
        # through its int16 bit pattern) yields a tensor whose storage is exactly the slice, with no 5 MB copy.
        def own(t):
            if output_dir is None or not t.is_contiguous() or t.numel() == 0:
                return t.clone() if output_dir is not None else t
            if t.dtype == torch.bfloat16:
                return torch.from_numpy(t.view(torch.int16).numpy()).view(torch.bfloat16)
            return torch.from_numpy(t.numpy())
        for row in (range(len(ids)) if rows is None else rows):
            seq_id, label = ids[row], labels[row]
            n = min(truncation_seq_length, len(strs[row]))
            result = {"label": label}
            if "per_tok" in include:
                result["representations"] = {l: own(t[row, 1:n + 1]) for l, t in reps.items()}
            means = {l: means_b[l][row] for l in reps}
            if "mean" in include:
                result["mean_representations"] = {l: m.clone() for l, m in means.items()}
            if "bos" in include:
                result["bos_representations"] = {l: own(t[row, 0]) for l, t in reps.items()}
            if contacts is not None:
                result["contacts"] = own(contacts[row, :n, :n])
            if output_dir is not None:
                path = output_dir / f"{label}.pt"
                path.parent.mkdir(parents=True, exist_ok=True)
                torch.save(result, path)
            rows_idx.append(seq_id)
            for l in layers:
                rows_mean[l].append(means[l])
        with lock:
            my_index.extend(rows_idx)
            for l in layers:
                my_means[l].extend(rows_mean[l])

    on_gpu = device is not None and device.type == "cuda"
    use_writer = on_gpu or async_host
    n_threads = writer_threads or default_writer_threads()
    writer = _Writer(depth=writer_depth, threads=n_threads) if use_writer else None
    pool = _PinnedPool() if on_gpu else None
    copy_stream = torch.cuda.Stream(device) if on_gpu else None
    with torch.no_grad():
        for n_done, bid in enumerate(plan[rank]):
            ids = batches[bid]
            labels, strs, toks = convert([dataset[i] for i in ids])
            log(f"[rank {rank}] batch {n_done + 1}/{len(plan[rank])}: {toks.size(0)} sequences x {toks.size(1)} tokens")
            kw = {}
            if getattr(embed_fn, "wants_lengths", False):  # token counts, read before the tokens leave the host
                kw["lengths"] = (toks.ne(alphabet.padding_idx) * torch.arange(1, toks.size(1) + 1)).amax(dim=1)
            if device is not None:
                toks = toks.to(device=device, non_blocking=True)
            out = embed_fn(toks, layers, return_contacts, **kw)
            reps = {l: t for l, t in out["representations"].items()}
            contacts = out["contacts"] if return_contacts else None
            means_dev = batch_means(reps, strs)
            if not use_writer:
                finish(ids, labels, strs, {l: t.to("cpu") for l, t in reps.items()},
                       contacts.to("cpu") if contacts is not None else None, {l: m.to("cpu") for l, m in means_dev.items()})
                continue
            need_full = ("per_tok" in include) or ("bos" in include)
            host_reps, host_contacts, host_means, copied = {}, None, {}, None
            if on_gpu:
                # device -> pinned host on the copy stream, results handled by the writer threads
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(device))
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(ready)
                    for l, t in reps.items():
                        if not need_full:  # only the [B,E] means leave the device
                            t = t[:, :1, :]
                        host_reps[l] = pool.take(t)
                        host_reps[l].copy_(t, non_blocking=True)
                        host_means[l] = pool.take(means_dev[l])
                        host_means[l].copy_(means_dev[l], non_blocking=True)
                    if contacts is not None:
                        host_contacts = pool.take(contacts)
                        host_contacts.copy_(contacts, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record(copy_stream)
            else:  # async_host: the stub's CPU tensors are the "host copies"
                host_reps = {l: t.contiguous() for l, t in reps.items()}
                host_contacts = contacts
                host_means = dict(means_dev)

            def chunk_job(rows, ids=ids, labels=labels, strs=strs, host_reps=host_reps, host_contacts=host_contacts,
                          copied=copied, host_means=host_means):
                if copied is not None:
                    copied.synchronize()
                finish(ids, labels, strs, host_reps, host_contacts, {l: m.clone() for l, m in host_means.items()}, rows)

            def done(host_reps=host_reps, host_contacts=host_contacts, host_means=host_means,
                     keep_alive=(reps, contacts, means_dev)):
                if pool is None:
                    return
                for t in list(host_reps.values()) + list(host_means.values()):
                    pool.give(t)
                if host_contacts is not None:
                    pool.give(host_contacts)

            step = max(1, int(chunk_rows))
            # (chunk_job is bound NOW: the name is re-bound by the next batch before the writer threads run)
            writer.submit([(lambda r=range(a, min(a + step, len(ids))), f=chunk_job: f(r)) for a in range(0, len(ids), step)], done)
    if writer is not None:
        writer.close()
    gathered: Dict[int, torch.Tensor] = {}
    if gather_mean:
        dev = device if device is not None else torch.device("cpu")
        idx = torch.tensor(my_index, dtype=torch.int64, device=dev)
        for l in layers:
            rows = (torch.stack(my_means[l]).float() if my_means[l] else torch.zeros((0, embed_dim))).to(dev)
            gathered[l] = gather_rows(rows.reshape(-1, embed_dim), idx, len(dataset))
    return gathered


def make_embed_fn(model, varlen: bool = True):
    """The ``embed_fn`` of :func:`extract` for an engine model: token-packed batches (no compute on padding)
    where the model has them and no contact maps are asked for (those are per-sequence [T,T]: padded path)."""
    varlen = varlen and getattr(model, "supports_varlen", False)

    def embed_fn(toks, layers, return_contacts, lengths=None):
        if varlen and not return_contacts:
            return model.forward_varlen(toks, repr_layers=layers, lengths=lengths)
        if return_contacts and getattr(model, "supports_contacts_only", False):
            # only the map is kept (scripts/extract.py:104-131 never looks at "attentions" / "logits")
            return model(toks, repr_layers=layers, contacts_only=True)
        return model(toks, repr_layers=layers, return_contacts=return_contacts)

    embed_fn.wants_lengths = True
    return embed_fn


def create_parser():
    p = argparse.ArgumentParser(description="Sharded per-token / mean representation extraction on MI355X GPUs")
    p.add_argument("model_location", type=str)
    p.add_argument("fasta_file", type=pathlib.Path)
    p.add_argument("output_dir", type=pathlib.Path)
    p.add_argument("--toks_per_batch", type=int, default=65536,
                   help="maximum batch size in tokens (the reference's scripts/extract.py defaults to 4096, sized for "
                        "a 16 GB GPU; 64 k tokens give every GEMM of the layer whole rounds of 256x256 tiles on the 256 "
                        "CUs: 615 k residues/s vs 351 k at 4096)")
    p.add_argument("--repr_layers", type=int, default=[-1], nargs="+")
    p.add_argument("--include", type=str, nargs="+", choices=["mean", "per_tok", "bos", "contacts"], required=True)
    p.add_argument("--truncation_seq_length", type=int, default=1022)
    p.add_argument("--gpus", type=int, default=1,
                   help="GPUs of this node to shard the FASTA over; from a plain shell the command re-executes itself "
                        "as that many ranks (torch.distributed.run, RCCL), under a launcher it joins the existing ranks")
    p.add_argument("--nogpu", action="store_true",
                   help="accepted for command-line compatibility with scripts/extract.py and refused: the engine has "
                        "no CPU path")
    p.add_argument("--writer_threads", type=int, default=0, help="result-file writer threads (0 = from the host's cores)")
    p.add_argument("--no_crc32", action="store_true",
                   help="write the result files WITHOUT the zip CRC32 of their records (torch.serialization.set_crc32_options): "
                        "torch.load never checks it, and the checksum is 1.2 of the 3.3 ms a 5 MB per-token file costs (+ 6 %% "
                        "per-token extraction at 8 ranks per host), but zipfile / unzip -t and archive tooling report such files "
                        "as corrupt.  Default: files like the reference's scripts/extract.py writes them, checksum included; "
                        "ESM_AMD_EXTRACT_NO_CRC32=1 is the same switch")
    p.add_argument("--crc32", action="store_true", help=argparse.SUPPRESS)  # round-4 spelling of the (now default) behaviour
    p.add_argument("--no_varlen", action="store_true",
                   help="always run padded batches (default: token-packed batches whenever they save >= 8 %% of the rows)")
    p.add_argument("--mean_matrix", type=pathlib.Path, default=None,
                   help="rank 0 also writes {layer: [n_sequences, E]} gathered mean embeddings to this file")
    return p


def main(argv=None):
    args = create_parser().parse_args(argv)
    from . import FastaBatchedDataset, pretrained
    from .launch import init_ranks, relaunch, under_launcher
    from .msa_transformer import MSATransformer

    if args.nogpu:
        raise SystemExit("esm_amd.extract: --nogpu is not available — the forward pass exists only as gfx950 kernels "
                         "(use the reference's scripts/extract.py for a CPU run)")

    if args.gpus > 1 and not under_launcher():  # plain shell: become N ranks, one per GPU
        raise SystemExit(relaunch(args.gpus, ("-m", "esm_amd.extract"), sys.argv[1:] if argv is None else argv))
    if not torch.cuda.is_available():
        raise RuntimeError("esm_amd.extract needs an MI355X: the engine has no CPU fallback")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    dist, rank, world, local_rank = init_ranks(world_env, "nccl")  # nccl == RCCL on ROCm
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    no_crc = (args.no_crc32 or os.environ.get("ESM_AMD_EXTRACT_NO_CRC32", "0") == "1") and not args.crc32
    if no_crc and hasattr(torch.serialization, "set_crc32_options"):
        torch.serialization.set_crc32_options(False)  # opt-in, process wide: this process only writes result files
    model, alphabet = pretrained.load_model_and_alphabet(args.model_location)
    if isinstance(model, MSATransformer):  # scripts/extract.py:66-69
        raise ValueError("This script currently does not handle models with MSA input (MSA Transformer).")
    model = model.eval().to(dev)
    dataset = FastaBatchedDataset.from_file(args.fasta_file)
    if rank == 0:
        print(f"Read {args.fasta_file} with {len(dataset)} sequences; {world} rank(s)")

    embed_fn = make_embed_fn(model, varlen=not args.no_varlen)

    means = extract(dataset, alphabet, embed_fn, model.num_layers, model.embed_dim, args.repr_layers, args.include,
                    output_dir=args.output_dir, toks_per_batch=args.toks_per_batch,
                    truncation_seq_length=args.truncation_seq_length, device=dev,
                    gather_mean=args.mean_matrix is not None, writer_threads=args.writer_threads)
    if args.mean_matrix is not None and rank == 0:
        torch.save({"labels": dataset.sequence_labels, "mean_representations": {l: t.cpu() for l, t in means.items()}},
                   args.mean_matrix)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
