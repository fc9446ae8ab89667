"""Small-batch GEMMs with HOT weights (the same matrix every launch: what an isolated micro-benchmark measures) against
COLD ones (a different matrix every launch, 33+ of them: what a forward pass sees — the 650M model's 1.3 GB of fp16
weights do not fit the 256 MiB MALL), optionally with the next launch's weights touched on a side stream while the
current launch runs (a software prefetch into the memory-side cache).

    python tools/bench_cold_weights.py [--B 4]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import _native as nat  # noqa: E402
from esm_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--nw", type=int, default=40)
    args = ap.parse_args()
    M, E, F = args.B * 1024, 1280, 5120
    g = torch.Generator(device="cuda").manual_seed(0)
    side = torch.cuda.Stream()
    for name, N, K, epi in (("v / out", E, E, nat.EPI_STORE_T), ("fc1 gelu", F, E, nat.EPI_GELU_T), ("fc2 resid", E, F, nat.EPI_RESID_F32)):
        a = torch.randn(M, K, device="cuda", generator=g).half()
        ws = [(torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).half() for _ in range(args.nw)]
        bias = torch.randn(N, device="cuda", generator=g)
        out = torch.zeros(M, N, device="cuda") if epi == nat.EPI_RESID_F32 else None
        filler = torch.randn(64 << 20, device="cuda", generator=g)  # 256 MiB: what the rest of a layer streams in between

        def run(mode, reps=3):
            best = 1e9
            for _ in range(reps):
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.nw)]
                for i in range(args.nw):
                    w = ws[0] if mode == "hot" else ws[i]
                    if mode == "cold":
                        filler.add_(1.0)  # evicts the MALL between launches (untimed)
                    if mode == "prefetch" and i + 1 < args.nw:
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            ws[i + 1].view(torch.int32)[:, ::32].sum()  # one dword per 128-byte line
                    ev[2 * i].record()
                    ops.linear(a, w, bias, epi, out=out)
                    ev[2 * i + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(2, args.nw))
                best = min(best, ts[len(ts) // 2])
            return best * 1e3

        print(f"B={args.B} {name:10s} M={M} N={N} K={K}: hot {run('hot'):7.1f} us | rotating {run('rot'):7.1f} us | "
              f"rotating + MALL flushed {run('cold'):7.1f} us | rotating + next weights touched on a side stream {run('prefetch'):7.1f} us",
              flush=True)
        del ws, filler


if __name__ == "__main__":
    main()
