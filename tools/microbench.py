"""Kernel micro-benchmarks on the bench shapes (650M, B sequences x 1024 tokens), timed with
HIP events on the launch stream.   python tools/microbench.py [--B 64] [--iters 10] [--only gemm]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import ops, _native as nat


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--panels", action="store_true", help="sweep the tile-order panel width of the persistent GEMM")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    T, E, H, F = 1024, 1280, 20, 5120
    M = args.B * T
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    if not args.only or "gemm" in args.only:
        for name, N, K, epi in [("qkv-like store", 2 * E, E, nat.EPI_STORE_T), ("out_proj resid", E, E, nat.EPI_RESID_F32),
                                ("fc1 gelu", F, E, nat.EPI_GELU_T), ("fc1 store", F, E, nat.EPI_STORE_T),
                                ("fc2 resid", E, F, nat.EPI_RESID_F32), ("fc2 store", E, F, nat.EPI_STORE_T)]:
            a = rnd(M, K).to(dt); w = (rnd(N, K) / math.sqrt(K)).to(dt); bias = rnd(N)
            out = torch.zeros(M, N, device="cuda") if epi == nat.EPI_RESID_F32 else None
            variants = [("persistent", dict()), ("tile-kernel", dict(force_old=True))]
            if args.panels:
                variants += [(f"persistent C={c}", dict(panel_c=c)) for c in (1, 2, 4, 5, 10, 20) if c <= (N + 255) // 256]
            for vname, kw in variants:
                ms = timeit(lambda: ops.linear(a, w, bias, epi, out=out, **kw), args.iters)
                print(f"gemm {name:16s} {vname:18s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
            del a, w, out
    if "dbg" in args.only:
        N, K = E, F
        a = rnd(M, K).to(dt); w = (rnd(N, K) / math.sqrt(K)).to(dt); bias = rnd(N)
        for dbg, what in [(0, "full 2+1+1+0"), (1, "no staging"), (3, "no staging, no barrier"), (2, "staging, no wait/barrier"),
                          (5, "MFMA + barrier only"), (7, "MFMA only"), (8, "staging only"), (24, "staging only, linear src"),
                          (16, "full, linear src"), (40, "staging only, L2-resident slab"), (32, "full 2+1+1, L2-resident slab"),
                          (64, "full, burst 4+0+0+0"), (128, "full, 2+2+0+0"), (96, "burst, L2-resident"), (160, "2+2, L2-resident")]:
            ms = timeit(lambda: ops.linear(a, w, bias, nat.EPI_STORE_T, dbg=dbg), args.iters)
            print(f"fc2-shape dbg={dbg} ({what}): {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
    if not args.only or "qkv" in args.only:
        hnd = ops.QkvHandle(E, H, dt)
        a = rnd(M, E).to(dt); w = (rnd(3 * E, E) / math.sqrt(E)).to(dt); bias = rnd(3 * E)
        ms = timeit(lambda: hnd(a, w, bias, args.B, T), args.iters)
        print(f"gemm qkv+rope       M={M} N={3*E} K={E}: {ms*1e3:8.1f} us  {2*M*3*E*E/ms/1e9:7.1f} TFLOP/s", flush=True)
    if not args.only or "attn" in args.only:
        q = (rnd(args.B, H, T, 64) * 0.5).to(dt); k = (rnd(args.B, H, T, 64) * 0.5).to(dt); v = rnd(args.B, H, T, 64).to(dt)
        vt = ops.make_vt(v)
        ms = timeit(lambda: ops.attention(q, k, vt), args.iters)
        print(f"attention B={args.B} H={H} T={T}: {ms*1e3:8.1f} us  {4*args.B*H*T*T*64/ms/1e9:7.1f} TFLOP/s", flush=True)
    if not args.only or "ln" in args.only:
        x = rnd(M, E); gm = rnd(E); bt = rnd(E)
        ms = timeit(lambda: ops.layernorm(x, gm, bt, dt), args.iters)
        print(f"layernorm rows={M} E={E}: {ms*1e3:8.1f} us  {M*E*6/ms/1e6:7.1f} GB/s (6E B/row)", flush=True)


if __name__ == "__main__":
    main()
