"""Small-batch study (not product code): the residual GEMMs of a 650M layer at B = 4 sequences (M = 4096 rows) have
80 output tiles for 256 CUs.  Times the fused residual epilogue against S fp32 K-slices launched as one batched
persistent GEMM (esmk_debug_linear_splitk) and checks that the slices sum to the same product.

    python tools/bench_splitk.py [--M 4096] [--iters 20]"""
import argparse
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import _native as nat  # noqa: E402
from esm_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    M = args.M
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for name, N, K, splits in (("fc2", 1280, 5120, (2, 4, 5, 8)), ("out_proj", 1280, 1280, (2, 4, 5))):
        a = rnd(M, K).half()
        w = (rnd(N, K) / math.sqrt(K)).half()
        bias = rnd(N)
        x = torch.zeros(M, N, device="cuda")
        us = timeit(lambda: ops.linear(a, w, bias, nat.EPI_RESID_F32, out=x), args.iters)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"{name:9s} M={M} N={N} K={K}: fused residual epilogue {us:7.1f} us  ({tiles} tiles)", flush=True)
        ref = a.float() @ w.float().t()
        for S in splits:
            part = torch.empty(S, M, N, device="cuda")
            fn = lambda: nat.check(nat.lib.esmk_debug_linear_splitk(
                nat.ptr(a), nat.ptr(w), nat.ptr(part), M, N, K, S, nat.dtype_code(a.dtype), nat.cur_stream()))
            us_s = timeit(fn, args.iters)
            err = (part.sum(0) - ref).abs().max().item() / ref.abs().max().item()
            # the consumer (LayerNorm) would read S extra fp32 rows: price them at the LayerNorm's measured rate
            extra = S * M * N * 4 / 4.0e12 * 1e6
            print(f"    S={S}: {us_s:7.1f} us ({tiles * S} tiles)  + ~{extra:4.1f} us to read the slices back; "
                  f"rel err of the sum {err:.1e}", flush=True)
        ln = timeit(lambda: ops.layernorm(x, bias, bias, torch.float16), args.iters)
        print(f"    (LayerNorm of the [{M},{N}] stream alone: {ln:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
