#!/usr/bin/env python
"""Forward latency / throughput of the ESM-2 engine over a list of (batch, length) shapes, with the engine's
per-kernel-class HIP-event breakdown.  Not the headline bench (bench.py); used to look at small-batch
(single-sequence, ESMFold-front-end-like) behaviour.

  python tools/bench_shapes.py --model 650M --shapes 1x1022,1x256,4x1022,16x1022,64x1022
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
    ap.add_argument("--shapes", default="1x1022,1x256,4x1022,16x1022,64x1022")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--classes", action="store_true", help="print the per-kernel-class breakdown")
    args = ap.parse_args()
    name = next(k for k in ESM2_DIMS if k == args.model or k.split("_")[2] == args.model)
    L, E, H = ESM2_DIMS[name]
    model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0))
    model = model.cuda()
    for shp in args.shapes.split(","):
        B, T = (int(v) for v in shp.split("x"))
        toks = synth_tokens(B, T, seed=1).cuda()
        with torch.no_grad():
            for _ in range(3):
                model(toks, repr_layers=[L])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                model(toks, repr_layers=[L])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
            model.profile_begin()
            model(toks, repr_layers=[L])
            prof = model.profile_end()
        print("B=%d T=%d  %.3f ms/forward  %.0f residues/s  (kernel sum %.3f ms)" % (
            B, T, dt * 1e3, B * T / dt, sum(e["ms"] for e in prof)), flush=True)
        if args.classes:
            for e in sorted(prof, key=lambda e: -e["ms"]):
                tf = e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["flops"] else 0.0
                print("    %-22s %4d launches %9.3f ms  %7.1f TF/s %8.1f GB/s" % (
                    e["name"], e["launches"], e["ms"], tf, e["bytes"] / (e["ms"] * 1e-3) / 1e9))


if __name__ == "__main__":
    main()
