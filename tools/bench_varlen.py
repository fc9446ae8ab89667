#!/usr/bin/env python
"""Padded vs token-packed forward on a mixed-length workload.

Sequence lengths are drawn from a log-normal fitted to UniRef50-like proteins (median ~270 residues, clipped to
[30, 1022]); batches are formed the way the reference does it (esm/data.py:get_batch_indices: sort by length,
fill up to toks_per_batch) or in file order (--unsorted; what a streaming service sees).  Reports REAL residues
per second (pad positions do not count) for model.forward and model.forward_varlen on the same batches.

  python tools/bench_varlen.py --model 650M --n 2048 --toks-per-batch 65536
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import esm  # noqa: E402
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict  # noqa: E402


def make_batches(lengths, toks_per_batch, sort):
    order = sorted(range(len(lengths)), key=lambda i: lengths[i]) if sort else list(range(len(lengths)))
    batches, cur, mx = [], [], 0
    for i in order:
        n = lengths[i] + 2
        if cur and max(mx, n) * (len(cur) + 1) > toks_per_batch:
            batches.append(cur)
            cur, mx = [], 0
        cur.append(i)
        mx = max(mx, n)
    if cur:
        batches.append(cur)
    return batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="650M")
    ap.add_argument("--n", type=int, default=2048, help="number of sequences")
    ap.add_argument("--toks-per-batch", type=int, default=65536)
    ap.add_argument("--unsorted", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    name = next(k for k in ESM2_DIMS if k == args.model or k.split("_")[2] == args.model)
    L, E, H = ESM2_DIMS[name]
    model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0))
    model = model.cuda()

    g = torch.Generator().manual_seed(args.seed)
    lengths = torch.exp(torch.randn(args.n, generator=g) * 0.7 + 5.6).clamp(30, 1022).long().tolist()
    batches = make_batches(lengths, args.toks_per_batch, not args.unsorted)
    toks = []
    for idx in batches:
        T = max(lengths[i] for i in idx) + 2
        t = torch.full((len(idx), T), 1, dtype=torch.int64)
        for r, i in enumerate(idx):
            n = lengths[i]
            t[r, 0] = 0
            t[r, 1:n + 1] = torch.randint(4, 24, (n,), generator=g)
            t[r, n + 1] = 2
        toks.append(t)
    real = sum(n + 2 for n in lengths)
    padded = sum(t.numel() for t in toks)
    print("%d sequences, %d batches, %d real tokens, %d padded (%.1f %% padding), batching %s" % (
        args.n, len(batches), real, padded, 100.0 * (padded - real) / padded, "file order" if args.unsorted else "sorted"))

    def run(fn):
        with torch.no_grad():
            fn(toks[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in toks:
                fn(t)
            torch.cuda.synchronize()
        return time.perf_counter() - t0

    t_pad = run(lambda t: model(t.cuda(), repr_layers=[L]))
    t_var = run(lambda t: model.forward_varlen(t, repr_layers=[L]))
    t_raw = run(lambda t: model.forward_varlen(t, repr_layers=[L], unpack=False))
    print("padded  forward        : %8.1f ms  %9.0f real residues/s" % (t_pad * 1e3, real / t_pad))
    print("packed  forward_varlen : %8.1f ms  %9.0f real residues/s  (%.2fx)" % (t_var * 1e3, real / t_var, t_pad / t_var))
    print("packed, no unpack      : %8.1f ms  %9.0f real residues/s  (%.2fx)" % (t_raw * 1e3, real / t_raw, t_pad / t_raw))


if __name__ == "__main__":
    main()
