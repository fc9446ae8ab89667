"""Full-height (256 x 256) vs half-height (128 x 256) tiles of the persistent GEMM on the layer shapes at small
batches (B sequences x 1024 tokens): the price of a half tile (HM_TILE_COST in gemm8.hip) and the gain per shape.
    python tools/bench_half_tiles.py [--B 1 4 16 64]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import ops, _native as nat
from tools.microbench import timeit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, nargs="+", default=[1, 4, 16, 64])
    ap.add_argument("--iters", type=int, default=20)
    a_ = ap.parse_args()
    E, F, dt = 1280, 5120, torch.float16
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for B in a_.B:
        M = B * 1024
        for name, N, K, epi in [("qk (store)", 2 * E, E, nat.EPI_STORE_T), ("v/out resid", E, E, nat.EPI_RESID_F32),
                                ("fc1 gelu", F, E, nat.EPI_GELU_T), ("fc2 resid", E, F, nat.EPI_RESID_F32)]:
            a = rnd(M, K).to(dt); w = (rnd(N, K) / math.sqrt(K)).to(dt); bias = rnd(N)
            out = torch.zeros(M, N, device="cuda") if epi == nat.EPI_RESID_F32 else None
            t = {}
            for hm in (-1, 1, 0):
                t[hm] = timeit(lambda: ops.linear(a, w, bias, epi, out=out, half_m=hm), a_.iters) * 1e3
            t256, t128 = ((M + 255) // 256) * ((N + 255) // 256), ((M + 127) // 128) * ((N + 255) // 256)
            r256, r128 = -(-t256 // 256), -(-t128 // 256)
            print(f"B={B:3d} {name:12s} M={M} N={N} K={K}: full {t[-1]:7.1f} us ({t256} tiles, {r256} rounds)  half {t[1]:7.1f} us "
                  f"({t128} tiles, {r128} rounds)  auto {t[0]:7.1f} us   half-tile cost {t[1] / r128 / (t[-1] / r256):.2f}", flush=True)


if __name__ == "__main__":
    main()
