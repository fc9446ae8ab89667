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
    if "dbg8" in args.only:
        # timing experiments on the persistent GEMM (results are wrong for most dbg codes)
        import ctypes
        def stamps(fn, ntiles, nk):
            buf = torch.zeros(256 * 32 * 4, dtype=torch.int64, device="cuda")
            nat.check(nat.lib.esmk_debug_gemm_timing(ctypes.c_void_p(buf.data_ptr())))
            fn(); torch.cuda.synchronize()
            nat.check(nat.lib.esmk_debug_gemm_timing(ctypes.c_void_p(0)))
            ntiles = max(1, ntiles)
            full = buf.view(256, 32, 4)[:, :ntiles, :].double().cpu()
            full = full[full[:, 0, 0] > 0]  # small problems: only the workgroups that had a tile
            t = full[:, :, :3]
            wall = (full[:, -1, 3] - full[:, 0, 3])  # 100 MHz ticks between the first and the last epilogue end
            cyc = (t[:, -1, 2] - t[:, 0, 2])
            ghz = (cyc / wall.clamp(min=1)).mean().item() * 0.1 if ntiles > 1 else float("nan")
            loop = (t[:, :, 1] - t[:, :, 0]).mean().item() / nk
            epi = (t[:, :, 2] - t[:, :, 1]).mean().item()
            gap = (t[:, 1:, 0] - t[:, :-1, 2]).mean().item() if ntiles > 1 else 0.0
            tot = (t[:, -1, 2] - t[:, 0, 0]).mean().item()
            first = (t[:, 0, 1] - t[:, 0, 0]).mean().item() / nk
            totmax = (t[:, -1, 2] - t[:, 0, 0]).max().item()
            return (f"cycles: {loop:7.1f}/K-tile (first tile {first:7.1f}), epilogue {epi:8.0f}, seam {gap:7.0f}, "
                    f"total mean {tot:9.0f} max {totmax:9.0f}, shader clock {ghz:5.2f} GHz")
        for name, N, K in [("fc2 shape", E, F), ("fc1 shape", F, E)]:
            a = rnd(M, K).to(dt); w = (rnd(N, K) / math.sqrt(K)).to(dt); bias = rnd(N)
            ntiles, nk = (M // 256) * (N // 256) // 256, K // 64
            for dbg, what in [(0, "full"), (0x60, "DMA re-reads K slab 0"), (0x08, "no epilogue"),
                              (0x68, "slab 0, no epilogue"), (0x20, "epilogue w/o global stores"),
                              (0x40, "young stores in flight"), (0x01, "no MFMA"), (0x02, "no LDS-DMA"), (0x04, "no fragment reads"),
                              (0x06, "no DMA, no reads"), (0x0e, "MFMA + barriers only")]:
                fn = lambda: ops.linear(a, w, bias, nat.EPI_STORE_T, dbg=dbg)
                ms = timeit(fn, args.iters)
                extra = stamps(fn, ntiles, nk) if dbg in (0, 0x60, 0x08, 0x0e) else ""
                print(f"gemm8 {name} dbg={dbg:#04x} ({what:30s}): {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s  {extra}", flush=True)
            del a, w
        for name, N, K, epi in [("fc1 gelu", F, E, nat.EPI_GELU_T), ("fc2 resid", E, F, nat.EPI_RESID_F32), ("out resid", E, E, nat.EPI_RESID_F32)]:
            a = rnd(M, K).to(dt); w = (rnd(N, K) / math.sqrt(K)).to(dt); bias = rnd(N)
            out = torch.zeros(M, N, device="cuda") if epi == nat.EPI_RESID_F32 else None
            ntiles, nk = (M // 256) * (N // 256) // 256, K // 64
            for dbg, what in ((0, "default"), (0x90, "no residual prefetch")):
                if dbg == 0x90 and epi != nat.EPI_RESID_F32:
                    continue
                fn = lambda: ops.linear(a, w, bias, epi, out=out, dbg=dbg)
                ms = timeit(fn, args.iters)
                extra = stamps(fn, ntiles, nk)
                print(f"gemm8 {name} {what:16s}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s  {extra}", flush=True)
            del a, w, out
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
        for var in (0, 1, 2, 3, 6, 7):
            ms = timeit(lambda: ops.layernorm(x, gm, bt, dt, variant=var), args.iters)
            print(f"layernorm variant {var} rows={M} E={E}: {ms*1e3:8.1f} us  {M*E*6/ms/1e6:7.1f} GB/s (6E B/row)", flush=True)


if __name__ == "__main__":
    main()
