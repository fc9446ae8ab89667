"""Randomised shape sweep of the persistent GEMM (esmk_op_linear -> gemm8.hip) against torch fp32 matmul on
the same fp16/bf16 operand values.   python tools/fuzz_gemm.py [--cases 200] [--seed 0]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import ops, _native as nat


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(a.seed)
    gd = torch.Generator(device="cuda").manual_seed(a.seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())
    worst = 0.0
    for case in range(a.cases):
        dt = (torch.float16, torch.bfloat16)[ri(0, 1)]
        K = 64 * ri(1, 24)
        N = 8 * ri(1, 400)
        M = ri(1, 3000) if ri(0, 3) else 256 * ri(1, 40)
        epi = (nat.EPI_STORE_T, nat.EPI_STORE_F32, nat.EPI_GELU_T, nat.EPI_GELU_F32, nat.EPI_RESID_F32)[ri(0, 4)]
        use_bias = bool(ri(0, 3))
        x = torch.randn(M, K, device="cuda", generator=gd).to(dt)
        w = (torch.randn(N, K, device="cuda", generator=gd) / math.sqrt(K)).to(dt)
        bias = torch.randn(N, device="cuda", generator=gd) if use_bias else None
        ref = x.float() @ w.float().t()
        if bias is not None:
            ref = ref + bias
        out = None
        if epi == nat.EPI_RESID_F32:
            res = torch.randn(M, N, device="cuda", generator=gd)
            out = res.clone()
            ref = ref + res
        if epi in (nat.EPI_GELU_T, nat.EPI_GELU_F32):
            ref = gelu(ref)
        hm = (0, 1, -1)[ri(0, 2)]  # tile height: decided from the tile count / forced 128 rows / forced 256 rows
        got = ops.linear(x, w, bias, epi, out=out, panel_c=ri(0, 6), half_m=hm)
        if hm != 0:  # the two tile heights must agree bit for bit (same MFMA sequence over K per element)
            out2 = res.clone() if epi == nat.EPI_RESID_F32 else None
            other = ops.linear(x, w, bias, epi, out=out2, half_m=-hm)
            if not torch.equal(other, got):
                print(f"FAIL case {case}: half / full tile heights differ: M={M} N={N} K={K} epi={epi} dt={dt}")
                sys.exit(1)
        eps = 2.0 ** -11 if dt == torch.float16 else 2.0 ** -8
        tol = 1e-4 * math.sqrt(K) + (eps * ref.abs().max().item() if got.dtype != torch.float32 else 0) + 1e-5
        err = (got.float() - ref).abs().max().item()
        worst = max(worst, err / tol)
        if not (err <= tol) or not torch.isfinite(got.float()).all():
            print(f"FAIL case {case}: M={M} N={N} K={K} epi={epi} dt={dt} bias={use_bias} err={err:.3e} tol={tol:.3e}")
            sys.exit(1)
    print(f"{a.cases} cases ok, worst err/tol = {worst:.3f}")


if __name__ == "__main__":
    main()
