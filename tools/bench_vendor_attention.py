"""Calibration, not product code: torch's scaled_dot_product_attention backends (AOTriton / CK flash kernels
shipped with the ROCm wheel) on the bench attention shape [B,20,1024,64], next to attn_fwd_kernel on the same
q, k, v.     python tools/bench_vendor_attention.py [--B 64] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    B, H, T, D = args.B, 20, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    flops = 4.0 * B * H * T * T * D
    from torch.nn.attention import SDPBackend, sdpa_kernel

    for dt in (torch.float16, torch.bfloat16):
        q, k, v = (rnd(B, H, T, D) * 0.5).to(dt), (rnd(B, H, T, D) * 0.5).to(dt), rnd(B, H, T, D).to(dt)
        # the engine's q arrives pre-scaled by d^-1/2 (QKV epilogue); give the vendor call the same math
        vt = ops.make_vt(v)
        ms = timeit(lambda: ops.attention(q, k, vt), args.iters)
        print(f"{str(dt)[6:]:9s} attn_fwd_kernel            : {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF", flush=True)
        for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION)):
            try:
                with sdpa_kernel(be):
                    fn = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1.0)
                    ms = timeit(fn, args.iters)
                print(f"{str(dt)[6:]:9s} torch SDPA {name:16s}: {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF", flush=True)
            except Exception as e:  # backend not built for gfx950 in this wheel
                print(f"{str(dt)[6:]:9s} torch SDPA {name:16s}: unavailable ({str(e).splitlines()[0][:100]})", flush=True)


if __name__ == "__main__":
    main()
