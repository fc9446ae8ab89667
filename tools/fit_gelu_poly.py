"""Coefficients of the polynomial GELU of the GEMM epilogues (esm_amd/csrc/common.h: gelu_fast / gelu_fast_x4).

    gelu(x) = x * Phi(x) ~= x * (0.5 + u Q(t)),   u = clamp(x, -c, c),   t = 2 u^2 / c^2 - 1

Q is a degree-n polynomial fitted to erf(u / sqrt 2) / (2 u) on [0, c] with weight u^2 (Lawson-iterated weighted
least squares -> near-minimax in the ABSOLUTE error of gelu), evaluated by Horner in fp32 FMAs.  Prints the
coefficients (highest power first, as ESMK_GELU_COEF lists them) and the error of the fp32 evaluation against float64
erf on a dense grid.      python tools/fit_gelu_poly.py [--clamp 4.75] [--degree 11]
"""
import argparse

import numpy as np
import scipy.special as sp

f32 = np.float32


def fit(c, n, iters=400):
    N = 6000
    t = np.cos(np.pi * (np.arange(N) + 0.5) / N)
    u = np.sqrt((t + 1) / 2) * c
    f = 0.5 * sp.erf(u / np.sqrt(2)) / u
    w = np.maximum(u, 1e-2) ** 2
    V = np.vander(t, n + 1, increasing=True)
    wt = np.ones(N)
    best = None
    for _ in range(iters):
        coef, *_ = np.linalg.lstsq(V * (w * wt)[:, None], f * w * wt, rcond=None)
        err = np.abs((V @ coef - f) * w)
        if best is None or err.max() < best[1]:
            best = (coef.copy(), err.max())
        wt = wt * (err / err.max() + 1e-4) ** 0.5
        wt /= wt.max()
    return best[0][::-1].astype(f32)  # highest power first


def fma(a, b, c):
    # one rounding: the product of two float32 is exact in float64
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(f32)


def gelu_poly_f32(x, coef, clamp):
    """The device function, operation by operation, in emulated fp32."""
    x = x.astype(f32)
    u = np.clip(x, f32(-clamp), f32(clamp))
    t = fma((u * f32(2 / (clamp * clamp))).astype(f32), u, -1.0)
    q = np.full_like(t, coef[0])
    for ck in coef[1:]:
        q = fma(q, t, ck)
    return (x * fma(u, q, 0.5)).astype(f32)


def report(coef, clamp):
    rng = np.random.default_rng(0)
    xs = np.concatenate([np.linspace(-16, 16, 4000001), rng.normal(size=2000000) * 1.5])
    ref = xs * 0.5 * (1 + sp.erf(xs / np.sqrt(2)))
    d = np.abs(gelu_poly_f32(xs, coef, clamp).astype(np.float64) - ref)
    inside = np.abs(xs) <= clamp
    return d[inside].max(), (d[~inside] / np.abs(xs[~inside])).max()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--clamp", type=float, default=4.75)
    ap.add_argument("--degree", type=int, default=11)
    a = ap.parse_args()
    coef = fit(a.clamp, a.degree)
    print("k2 = %.9ef" % f32(2 / (a.clamp * a.clamp)))
    print(", ".join("%.9ef" % v for v in coef))
    e_in, e_out = report(coef, a.clamp)
    print("max |gelu_poly - gelu_erf| = %.3e for |x| <= %.2f;  max relative to |x| beyond = %.3e" % (e_in, a.clamp, e_out))
