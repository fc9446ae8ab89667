#!/usr/bin/env python
"""Contact-map throughput: ``model.predict_contacts`` (map accumulated layer by layer, no attention tensor;
csrc/contacts.hip) against ``model(tokens, return_contacts=True)`` (materialises [B,L,H,T,T] like the reference).

    python tools/bench_contacts.py [--model 650M] [--len 1022] [--fused 4,16,64] [--materialised 4,16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import esm  # noqa: E402
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict, synth_tokens  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="650M")
    ap.add_argument("--len", type=int, default=1022)
    ap.add_argument("--fused", default="4,16,64")
    ap.add_argument("--materialised", default="4,16")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--layers", type=int, default=0, help="override the number of layers (profiling runs)")
    args = ap.parse_args()
    name = next(k for k in ESM2_DIMS if k == args.model or k.split("_")[2] == args.model)
    L, E, H = ESM2_DIMS[name]
    L = args.layers or L
    model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0))
    model = model.cuda()

    def run(label, B, fn):
        toks = synth_tokens(B, args.len, seed=1).cuda()
        with torch.no_grad():
            fn(toks)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                fn(toks)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
            model.profile_begin()
            fn(toks)
            prof = model.profile_end()
        peak = torch.cuda.max_memory_allocated() / 2**30
        tot = sum(e["ms"] for e in prof)
        part = {e["name"]: e["ms"] for e in prof if e["name"] in ("attention_probs", "contacts")}
        print(f"{label:13s} B={B:3d} L={args.len}: {dt*1e3:9.2f} ms  {B*args.len/dt:10.0f} residues/s  peak {peak:6.1f} GiB  "
              f"(kernels {tot:.1f} ms, of which " + ", ".join(f"{k} {v:.1f}" for k, v in part.items()) + ")", flush=True)
        del toks
        model._engine.workspace = None
        torch.cuda.empty_cache()

    for B in [int(v) for v in args.fused.split(",") if v]:
        run("fused", B, lambda t: model.predict_contacts(t))
    for B in [int(v) for v in args.materialised.split(",") if v]:
        run("materialised", B, lambda t: model(t, return_contacts=True)["contacts"])


if __name__ == "__main__":
    main()
