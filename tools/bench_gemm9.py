"""gemm9 (one wave per SIMD, 128 x 128 wave blocks) against gemm8 (two waves per SIMD) and the vendor library on the
SAME operands, inside one process, interleaved rounds (boxes differ by +-3 %, so variants are only compared in-call).

    python tools/bench_gemm9.py [--B 64] [--rounds 5] [--iters 10] [--check-only] [--no-vendor]

1. correctness: gemm9 must be BIT-IDENTICAL to gemm8 (same MFMA sequence per output element) on ragged shapes, all
   linear epilogues, fp16 and bf16;
2. timing: per layer shape and epilogue, median over rounds of HIP-event timed loops, + in-kernel cycle stamps
   (cycles per K tile, epilogue, seam, effective shader clock).
"""
import argparse
import ctypes
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import _native as nat  # noqa: E402
from esm_amd import ops  # noqa: E402


def set_impl(impl, var=0, desync=0.0, group=0):
    nat.check(nat.lib.esmk_debug_gemm_impl(impl, var))
    nat.check(nat.lib.esmk_debug_set(b"resid_desync", float(desync)))
    nat.check(nat.lib.esmk_debug_set(b"resid_desync_group", float(group)))


def linear_ln_producer(a, w, bias, out, h16, part, mean, half_m=0):
    """the LayerNorm-fold producer form of the residual GEMM (esmk_op_linear_ln, epilogue 4)"""
    M, K = a.shape
    Nn = w.shape[0]
    nat.check(nat.lib.esmk_op_linear_ln(nat.ptr(a), nat.ptr(w), nat.ptr(bias), nat.ptr(None), nat.ptr(out), M, Nn, K, nat.EPI_RESID_F32,
                                        nat.dtype_code(a.dtype), nat.ptr(None), nat.ptr(h16), h16.shape[1], nat.ptr(part), part.shape[1],
                                        nat.ptr(mean), half_m, nat.cur_stream()))
    return out


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


def stamps(fn, ntiles, nk):
    buf = torch.zeros(256 * 32 * 4, dtype=torch.int64, device="cuda")
    nat.check(nat.lib.esmk_debug_gemm_timing(ctypes.c_void_p(buf.data_ptr())))
    fn()
    torch.cuda.synchronize()
    nat.check(nat.lib.esmk_debug_gemm_timing(ctypes.c_void_p(0)))
    full = buf.view(256, 32, 4)[:, :max(1, ntiles), :].double().cpu()
    full = full[full[:, 0, 0] > 0]
    t = full[:, :, :3]
    wall = full[:, -1, 3] - full[:, 0, 3]  # 100 MHz ticks
    cyc = t[:, -1, 2] - t[:, 0, 2]
    ghz = (cyc / wall.clamp(min=1)).mean().item() * 0.1 if ntiles > 1 else float("nan")
    loop = (t[:, :, 1] - t[:, :, 0]).mean().item() / nk
    epi = (t[:, :, 2] - t[:, :, 1]).mean().item()
    seam = (t[:, 1:, 0] - t[:, :-1, 2]).mean().item() if ntiles > 1 else float("nan")
    return loop, epi, seam, ghz


EXACT = True  # --tolerant: gemm9 and gemm8 only have to agree up to the summation order


def check(dt):
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    bad = 0
    shapes = [(256, 256, 64), (512, 768, 128), (1000, 1288, 192), (4096, 1280, 1280), (777, 264, 320), (2560, 5120, 1280),
              (70000, 1280, 1280), (300, 8, 64)]
    for M, N, K in shapes:
        a = rnd(M, K).to(dt)
        w = (rnd(N, K) / math.sqrt(K)).to(dt)
        bias = rnd(N)
        for epi in (nat.EPI_STORE_T, nat.EPI_STORE_F32, nat.EPI_GELU_T, nat.EPI_GELU_F32, nat.EPI_RESID_F32):
            for use_bias in (True, False):
                outs = []
                for impl, var in ((8, 0), (9, 0)):
                    set_impl(impl, var)
                    x0 = rnd(M, N) if epi == nat.EPI_RESID_F32 else None
                    if x0 is not None:
                        torch.manual_seed(0)
                        x0 = torch.arange(M * N, device="cuda", dtype=torch.float32).reshape(M, N).sin()
                    outs.append(ops.linear(a, w, bias if use_bias else None, epi, out=x0).clone())
                set_impl(8)
                ref = torch.nn.functional.linear(a.float(), w.float(), bias if use_bias else None)
                for name, o in (("gemm9", outs[1]),):
                    if EXACT:
                        same = torch.equal(o, outs[0])
                    else:  # different MFMA shapes in the two kernels: same value up to the summation order
                        tol = 2e-3 if o.dtype != torch.float32 else 2e-6
                        scale = (ref.abs().max().item() + (x0.abs().max().item() if x0 is not None else 0.0))
                        same = (o.float() - outs[0].float()).abs().max().item() <= tol * scale
                    fin = torch.isfinite(o.float()).all().item()
                    if not (same and fin):
                        bad += 1
                        d = (o.float() - outs[0].float()).abs().max().item()
                        print(f"MISMATCH {name} dtype={dt} M={M} N={N} K={K} epi={epi} bias={use_bias}: max|d|={d:.3e} finite={fin}",
                              flush=True)
                if epi == nat.EPI_STORE_F32:  # and gemm8 itself against fp32 torch (sanity of the reference arm)
                    e = (outs[0] - ref).abs().max().item() / ref.abs().max().item()
                    assert e < 5e-3, e
    print(f"check {dt}: {'OK (' + ('bit-identical to' if EXACT else 'within the summation-order tolerance of') + ' gemm8 on all shapes / epilogues)' if bad == 0 else str(bad) + ' MISMATCHES'}", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--no-vendor", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--tolerant", action="store_true", help="gemm9 == gemm8 up to the fp32 summation order instead of bit for bit")
    ap.add_argument("--half", action="store_true", help="time the half-height kernels (small batches) and their experiment variants")
    ap.add_argument("--cases", default="", help="comma-separated substrings of case names to run (default: all)")
    ap.add_argument("--dbg", action="store_true", help="also the timing-experiment variants of gemm9 (plain store only)")
    ap.add_argument("--desync", default="", help="residual epilogues: comma-separated fraction:group arms, e.g. 0.25:0,0.5:0,0.5:1,0.5:2")
    args = ap.parse_args()
    print("device:", torch.cuda.get_device_name(0), flush=True)
    global EXACT
    EXACT = not args.tolerant
    bad = 0
    # fp16 subnormal operands through the MFMA (the split-weight precision mode stores W - fp16(W), mostly subnormal)
    a1 = torch.ones(256, 64, device="cuda", dtype=torch.float16)
    w1 = torch.full((256, 64), 2.0 ** -20, device="cuda", dtype=torch.float16)
    o1 = ops.linear(a1, w1, None, nat.EPI_STORE_F32)
    print(f"fp16 subnormal operand through the MFMA: 64 * 2^-20 -> {o1[0, 0].item():.6e} (exact 6.103516e-05; 0 = flushed)", flush=True)
    if not args.no_check:
        bad = check(torch.float16) + check(torch.bfloat16)
    if args.check_only:
        sys.exit(1 if bad else 0)
    T, E, F = 1024, 1280, 5120
    M = args.B * T
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    dt = torch.float16
    cases = [("qk store", 2 * E, E, nat.EPI_STORE_T), ("v/out store", E, E, nat.EPI_STORE_T), ("out resid", E, E, nat.EPI_RESID_F32),
             ("fc1 store", F, E, nat.EPI_STORE_T), ("fc1 gelu", F, E, nat.EPI_GELU_T), ("fc2 store", E, F, nat.EPI_STORE_T),
             ("fc2 resid", E, F, nat.EPI_RESID_F32), ("out store32", E, E, nat.EPI_STORE_F32), ("fc2 store32", E, F, nat.EPI_STORE_F32),
             ("out gelu32", E, E, nat.EPI_GELU_F32), ("fc2 gelu32", E, F, nat.EPI_GELU_F32)]
    if args.cases:
        cases = [c for c in cases if any(k in c[0] for k in args.cases.split(","))]
    for name, N, K, epi in cases:
        a = rnd(M, K).to(dt)
        w = (rnd(N, K) / math.sqrt(K)).to(dt)
        bias = rnd(N)
        out = torch.zeros(M, N, device="cuda") if epi == nat.EPI_RESID_F32 else None
        flops = 2.0 * M * N * K
        arms = [("gemm8", 8, 0), ("gemm9", 9, 0)]
        hm = 0
        if args.half:
            hm = 1
            arms = [("gemm8 half", 8, 0), ("gemm9 half", 9, 0)]
            if epi == nat.EPI_STORE_T:
                arms += [("g9h no-barrier", 9, 8), ("g9h no-mfma", 9, 16), ("g9h no-dma", 9, 32), ("g9h no-reads", 9, 64), ("g9h no-epi", 9, 128),
                         ("g9h mfma-only", 9, 96), ("g9h skeleton", 9, 224)]
        if args.dbg and epi == nat.EPI_STORE_T and not args.half:
            arms += [("g9 dense", 9, 1), ("g9 spread24-61", 9, 2), ("g9 24+2k", 9, 3), ("g9 temporal-st", 9, 4096), ("g9 no-barrier", 9, 8),
                     ("g9 no-mfma", 9, 16), ("g9 no-dma", 9, 32), ("g9 no-reads", 9, 64), ("g9 no-epi", 9, 128), ("g9 mfma-only", 9, 96)]
        arms = [a + (0.0, 0) for a in arms]
        if args.desync and epi == nat.EPI_RESID_F32 and not args.half:
            for spec in args.desync.split(","):
                frac, grp = spec.split(":")
                arms.append((f"g9 desync {frac} g{grp}", 9, 0, float(frac), int(grp)))
        times = {a[0]: [] for a in arms}
        if not args.no_vendor and epi == nat.EPI_STORE_T:
            times["vendor"] = []
            bias_t = bias.to(dt)
        for _ in range(args.rounds):
            for n, impl, var, dsf, dsg in arms:
                set_impl(impl, var, dsf, dsg)
                times[n].append(timeit(lambda: ops.linear(a, w, bias, epi, out=out, half_m=hm), args.iters))
            if "vendor" in times:
                times["vendor"].append(timeit(lambda: torch.nn.functional.linear(a, w, bias_t), args.iters))
        if args.desync and epi == nat.EPI_RESID_F32 and not args.half:  # a delayed start changes no bit
            outs = []
            for n, impl, var, dsf, dsg in arms[1:]:
                set_impl(impl, var, dsf, dsg)
                x0 = torch.arange(M * N, device="cuda", dtype=torch.float32).reshape(M, N).sin()
                outs.append(ops.linear(a, w, bias, epi, out=x0))
            same = all(torch.equal(o, outs[0]) for o in outs[1:])
            print(f"{name:12s} desync arms bit-identical to the plain launch: {same}", flush=True)
            bad += 0 if same else 1
            del outs, x0
        if epi == nat.EPI_RESID_F32 and not args.half:  # the LayerNorm-fold producer next to the plain residual epilogue
            h16 = torch.zeros(M, N, dtype=dt, device="cuda")
            part = torch.zeros(M, (N + 127) // 128, 2, device="cuda")
            mean = torch.zeros(M, device="cuda")
            set_impl(9, 0)
            fn = lambda: linear_ln_producer(a, w, bias, out, h16, part, mean)
            for dbg, tag in ((0, "LN-fold producer"), (1, "  no h16 stores"), (2, "  no statistics"), (4, "  no mean loads"), (7, "  none of them")):
                if nat.lib.esmk_debug_set(b"lnf_dbg", float(dbg)) != 0:
                    break  # the producer ablations exist only in -DESMK_EXPERIMENTS builds (common.h)
                ts = [timeit(fn, args.iters) for _ in range(args.rounds)]
                loop, ep, seam, ghz = stamps(fn, min(32, (((M + 255) // 256) * ((N + 255) // 256)) // 256), K // 64)
                ms = statistics.median(ts)
                print(f"{name:12s} {tag:18s} {ms*1e3:8.1f} us (min {min(ts)*1e3:8.1f}) {flops/ms/1e9:7.1f} TFLOP/s | cycles/K-tile {loop:7.1f} "
                      f"epilogue {ep:7.0f} seam {seam:6.0f} clock {ghz:4.2f} GHz", flush=True)
            nat.lib.esmk_debug_set(b"lnf_dbg", 0.0)
            del h16, part, mean
        nt = ((M + 255) // 256) * ((N + 255) // 256)
        for n, impl, var, dsf, dsg in arms:
            set_impl(impl, var, dsf, dsg)
            loop, ep, seam, ghz = stamps(lambda: ops.linear(a, w, bias, epi, out=out, half_m=hm), min(32, nt // 256), K // 64)
            ms = statistics.median(times[n])
            print(f"{name:12s} {n:18s} {ms*1e3:8.1f} us (min {min(times[n])*1e3:8.1f}) {flops/ms/1e9:7.1f} TFLOP/s | cycles/K-tile {loop:7.1f} "
                  f"epilogue {ep:7.0f} seam {seam:6.0f} clock {ghz:4.2f} GHz", flush=True)
        if "vendor" in times:
            ms = statistics.median(times["vendor"])
            print(f"{name:12s} {'vendor':18s} {ms*1e3:8.1f} us (min {min(times['vendor'])*1e3:8.1f}) {flops/ms/1e9:7.1f} TFLOP/s", flush=True)
        set_impl(8)
        del a, w, out
    if not args.cases and not args.half:
        # the fused q/k (scale + RoPE + head-major store) and v (transposed store) projections, one op = both launches
        qkv = ops.QkvHandle(E, E // 64, dt)
        a = rnd(M, E).to(dt)
        wq = (rnd(3 * E, E) / math.sqrt(E)).to(dt)
        bq = rnd(3 * E)
        flops = 2.0 * M * 3 * E * E
        res = {}
        for _ in range(args.rounds):
            for n, impl in (("gemm8", 8), ("gemm9", 9)):
                set_impl(impl, 0)
                res.setdefault(n, []).append(timeit(lambda: qkv(a, wq, bq, args.B, T), args.iters))
        set_impl(8)
        for n, ts in res.items():
            ms = statistics.median(ts)
            print(f"{'qkv fused':12s} {n:18s} {ms*1e3:8.1f} us (min {min(ts)*1e3:8.1f}) {flops/ms/1e9:7.1f} TFLOP/s  (q/k RoPE + v transposed, both launches, "
                  "output allocation included)", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
