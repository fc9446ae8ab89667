// gelu_poly.h — GELU of the GEMM epilogues as a packed-FMA polynomial (gfx950).  NOT compiled into libesmk.so in
// round 2: measured and held back, see README.md next to this file and DESIGN.md 4.1.  To enable: move the
// definitions below into esm_amd/csrc/common.h in place of gelu_fast and call gelu_poly_x4(v) on the four values of
// a register group at the EPI_GELU_T / EPI_GELU_F32 sites of gemm8.hip and gemm.hip (commit 68f7f64 is that change).
#pragma once
#define ESMK_DEV __device__ __forceinline__

// Same function as x * (0.5 + u Q(t)),  u = clamp(x, -4.75, 4.75),  t = 2 u^2 / 4.75^2 - 1,  Q a degree-11 minimax
// fit of erf(u / sqrt 2) / (2u) weighted by u^2 (tools/fit_gelu_poly.py regenerates the coefficients and re-checks the
// bound in emulated fp32): |gelu_poly - gelu_erf| <= 1.4e-6 for |x| <= 4.75 and <= 1.4e-6 |x| beyond, ~100x below the
// fp16 rounding of the value it feeds.  No rcp / exp (quarter rate) and nothing but FMAs, so that two elements share
// one v_pk_fma_f32: ~8 VALU issue cycles per element against ~24 for the erfc(|z|) = poly(t) exp(-z^2) form of
// round 1 (the GELU epilogue of fc1 is VALU bound: 10 k of the tile's 71 k cycles).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kGeluClamp = 4.75f, kGeluK2 = 8.864265680e-02f;  // 2 / 4.75^2
// Horner coefficients of Q, highest power of t first
#define ESMK_GELU_COEF                                                                                       \
    {-5.631324602e-04f, 1.756936894e-03f, -2.548059914e-03f, 4.025654402e-03f, -8.102229796e-03f, 1.411156729e-02f, \
     -2.135194838e-02f, 3.020246327e-02f, -4.060446471e-02f, 5.325455219e-02f, -7.366643846e-02f, 1.487480104e-01f}
ESMK_DEV float gelu_poly(float x) {
    constexpr float c[12] = ESMK_GELU_COEF;
    const float u = __builtin_amdgcn_fmed3f(x, -kGeluClamp, kGeluClamp);
    const float t = __builtin_fmaf(u * kGeluK2, u, -1.0f);
    float q = c[0];
#pragma unroll
    for (int k = 1; k < 12; ++k) q = __builtin_fmaf(q, t, c[k]);
    return x * __builtin_fmaf(u, q, 0.5f);
}
// In place on four consecutive values (float[4] or a 4-vector), two elements per instruction; bit-identical to
// gelu_poly on each element (same IEEE operations in the same order).  The two Horner chains are interleaved by
// hand: dependent packed FMAs need a wait state that the other chain fills.
template <typename V>
ESMK_DEV void gelu_poly_x4(V& v) {
    constexpr float c[12] = ESMK_GELU_COEF;
    const f32x2 xa = {v[0], v[1]}, xb = {v[2], v[3]};
    f32x2 ua, ub;
    ua.x = __builtin_amdgcn_fmed3f(xa.x, -kGeluClamp, kGeluClamp), ua.y = __builtin_amdgcn_fmed3f(xa.y, -kGeluClamp, kGeluClamp);
    ub.x = __builtin_amdgcn_fmed3f(xb.x, -kGeluClamp, kGeluClamp), ub.y = __builtin_amdgcn_fmed3f(xb.y, -kGeluClamp, kGeluClamp);
    const f32x2 ta = __builtin_elementwise_fma(ua * kGeluK2, ua, (f32x2)(-1.0f));
    const f32x2 tb = __builtin_elementwise_fma(ub * kGeluK2, ub, (f32x2)(-1.0f));
    f32x2 qa = (f32x2)(c[0]), qb = (f32x2)(c[0]);
#pragma unroll
    for (int k = 1; k < 12; ++k) {
        qa = __builtin_elementwise_fma(qa, ta, (f32x2)(c[k]));
        qb = __builtin_elementwise_fma(qb, tb, (f32x2)(c[k]));
    }
    const f32x2 ra = xa * __builtin_elementwise_fma(ua, qa, (f32x2)(0.5f));
    const f32x2 rb = xb * __builtin_elementwise_fma(ub, qb, (f32x2)(0.5f));
    v[0] = ra.x, v[1] = ra.y, v[2] = rb.x, v[3] = rb.y;
}

