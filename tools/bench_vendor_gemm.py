"""Calibration, not product code: the vendor library (hipBLASLt / rocBLAS through torch.nn.functional.linear)
on the five GEMM shapes of one ESM-2 650M layer at the bench batch, next to this repo's persistent kernel on the
SAME operands on the same box.  Answers "how far is gemm8 from what the best available gfx950 GEMM sustains under
the 1400 W cap" — the MFMA roof (2.5 PF) is a clock-times-width number no real-data kernel reaches.

    python tools/bench_vendor_gemm.py [--B 64] [--iters 20] [--zeros]

--zeros repeats every shape on zero-filled operands (no toggling: shows the clock/power effect on the same code).
Plain store epilogues on both sides (the vendor call has no fused GELU / RoPE / residual).
"""
import argparse
import math
import os
import subprocess
import sys
import threading
import time

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
    return a.elapsed_time(b) / iters


class Smi(threading.Thread):
    """Samples sclk / package power with rocm-smi while a loop runs."""

    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.samples = []

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "-c", "-P"], capture_output=True, text=True, timeout=10).stdout
                sclk = [l.split("(")[-1].split("Mhz")[0] for l in out.splitlines() if "sclk" in l]
                pw = [l.split(":")[-1].strip() for l in out.splitlines() if "Power (W)" in l]
                if sclk and pw:
                    self.samples.append((int(sclk[0]), float(pw[0])))
            except Exception:
                pass
            time.sleep(0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--smi", action="store_true", help="sample rocm-smi during a 3 s loop of each kernel on the fc1 shape")
    args = ap.parse_args()
    T, E, F = 1024, 1280, 5120
    M = args.B * T
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    print("device:", torch.cuda.get_device_name(0), "torch", torch.__version__, flush=True)
    for dt in (torch.float16, torch.bfloat16):
        for name, N, K in [("qk proj", 2 * E, E), ("v/out proj", E, E), ("fc1", F, E), ("fc2", E, F)]:
            a = rnd(M, K).to(dt)
            w = (rnd(N, K) / math.sqrt(K)).to(dt)
            bias = rnd(N)
            bias_t = bias.to(dt)
            out = torch.empty(M, N, device="cuda", dtype=dt)
            flops = 2.0 * M * N * K
            fv = lambda: torch.nn.functional.linear(a, w, bias_t)
            fo = lambda: ops.linear(a, w, bias, nat.EPI_STORE_T, out=out)
            nat.check(nat.lib.esmk_debug_gemm_impl(8, 0))
            msv, mso = timeit(fv, args.iters), timeit(fo, args.iters)
            nat.check(nat.lib.esmk_debug_gemm_impl(9, 0))
            ms9 = timeit(fo, args.iters)
            nat.check(nat.lib.esmk_debug_gemm_impl(8, 0))
            line = (f"{str(dt)[6:]:9s} {name:11s} M={M} N={N:5d} K={K:5d}: vendor {msv*1e3:8.1f} us {flops/msv/1e9:7.1f} TF | "
                    f"gemm8 {mso*1e3:8.1f} us {flops/mso/1e9:7.1f} TF | gemm9 {ms9*1e3:8.1f} us {flops/ms9/1e9:7.1f} TF")
            if args.zeros:
                a.zero_()
                w.zero_()
                zv, zo = timeit(fv, args.iters), timeit(fo, args.iters)
                line += f" | zero operands: vendor {flops/zv/1e9:7.1f} TF, gemm8 {flops/zo/1e9:7.1f} TF"
            print(line, flush=True)
            if args.smi and name == "fc1" and dt == torch.float16:
                a.copy_(rnd(M, K).to(dt))
                w.copy_((rnd(N, K) / math.sqrt(K)).to(dt))
                for label, fn in (("vendor", fv), ("gemm8", fo), ("gemm9", fo)):
                    nat.check(nat.lib.esmk_debug_gemm_impl(9 if label == "gemm9" else 8, 0))
                    s = Smi()
                    s.start()
                    t0 = time.time()
                    while time.time() - t0 < 3.0:
                        for _ in range(50):
                            fn()
                        torch.cuda.synchronize()
                    s.stop = True
                    s.join()
                    tail = s.samples[len(s.samples) // 3:]
                    if tail:
                        print(f"    rocm-smi under {label} loop: sclk {sum(x[0] for x in tail)/len(tail):.0f} MHz, "
                              f"power {sum(x[1] for x in tail)/len(tail):.0f} W ({len(tail)} samples)", flush=True)
            nat.check(nat.lib.esmk_debug_gemm_impl(0, 0))
            del a, w, out


if __name__ == "__main__":
    main()
