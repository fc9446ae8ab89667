"""Tile-order panel width of the persistent GEMM (how many column tiles an XCD's 32 concurrent tiles span) against time,
for gemm9 on the layer shapes: the loop is bound by the operand stream since round 3, so the L2 hit rate of the stream
may matter where it did not under the power cap (round-2 sweep on gemm8: within 0.5 %).

    python tools/sweep_panel_gemm9.py [--B 64]
"""
import argparse
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import _native as nat  # noqa: E402
from esm_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    M, E, F = args.B * 1024, 1280, 5120
    g = torch.Generator(device="cuda").manual_seed(0)
    nat.check(nat.lib.esmk_debug_gemm_impl(9, 0))
    try:
        for name, N, K, epi in (("qk", 2 * E, E, nat.EPI_STORE_T), ("v/out", E, E, nat.EPI_STORE_T), ("fc1 gelu", F, E, nat.EPI_GELU_T),
                                ("fc2 resid", E, F, nat.EPI_RESID_F32)):
            a = torch.randn(M, K, device="cuda", generator=g).half()
            w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).half()
            bias = torch.randn(N, device="cuda", generator=g)
            out = torch.zeros(M, N, device="cuda") if epi == nat.EPI_RESID_F32 else None
            tn = (N + 255) // 256
            widths = sorted({c for c in (0, 1, 2, 4, 5, 8, 10, 16, 20) if c <= tn})
            res = {c: [] for c in widths}
            for _ in range(args.rounds):
                for c in widths:
                    res[c].append(timeit(lambda: ops.linear(a, w, bias, epi, out=out, panel_c=c), 8))
            flops = 2.0 * M * N * K
            print(f"{name:10s} N={N} K={K} ({tn} column tiles): " + "  ".join(
                f"{'auto' if c == 0 else c}: {statistics.median(res[c]) * 1e3:.1f} us ({flops / statistics.median(res[c]) / 1e9:.0f} TF)" for c in widths), flush=True)
    finally:
        nat.check(nat.lib.esmk_debug_gemm_impl(0, 0))


if __name__ == "__main__":
    main()
