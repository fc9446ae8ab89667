// common.h — shared device helpers for the gfx950 (MI355X / CDNA4) ESM-2 engine.
// Written for gfx950 only: 64-lane wavefronts, v_mfma_f32_16x16x32_{f16,bf16} (linear layers: 13 % less energy per flop
// than the 32x32x16 shape under the package power cap, tools/mfma_power_probe.hip) and v_mfma_f32_32x32x16 (attention,
// contacts), LDS-DMA (global_load_lds_dwordx4 / buffer_load ... lds), 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ESMK_DEV __device__ __forceinline__

// Timing experiments that deliberately compute WRONG results (kernels with their MFMAs, exponentials, LDS-DMA, fragment
// reads or epilogue removed: gemm8 DBG / gemm9 VAR bits 8 .. 128 and 1024, ESMK_ATTN_HACK, the producer's lnf_dbg) exist
// only in builds made with ESMK_HIPCC_EXTRA="-DESMK_EXPERIMENTS" (esm_amd/build.py: part of the source hash, so such a
// library never passes for the shipped one).  In the shipped library every run-time switch leaves results unchanged.
#ifdef ESMK_EXPERIMENTS
constexpr bool kExperiments = true;
#else
constexpr bool kExperiments = false;
#endif

// Operand-type traits: T is the MFMA operand element (_Float16 or __bf16).
template <typename T>
struct Op;

template <>
struct Op<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static ESMK_DEV f32x16 mma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    // D = A.B + C with D in registers of its own: C (e.g. a bias broadcast that is reused for every tile) stays
    // intact.  The builtin lets the register allocator tie D to C and then copies C first (16 v_mov per tile).
    static ESMK_DEV f32x16 mma_keep_c(v8 a, v8 b, const f32x16& c) {
        f32x16 d;
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    // 16 x 16 x 32: D[i][j] += sum_k a[i][k] b[j][k];  lane l supplies a[l & 15][8 (l >> 4) .. + 8], b likewise, and
    // holds D[4 (l >> 4) + r][l & 15], r = 0..3
    static ESMK_DEV f32x4 mma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    // The same product accumulated IN PLACE, D tied to C in the AGPR file.  For the peeled first K tile of gemm9: left to the
    // builtin, the allocator gives the second K half's results registers of their own (D != C), runs out of AGPRs and parks
    // 24 - 40 accumulator quads in VGPRs (s_nop 7 + 4 v_accvgpr_read behind their MFMA, 4 v_accvgpr_write later).  The
    // compiler does not see an MFMA in here: the caller keeps readers of c (LDS writes, accvgpr reads) >= 12 wait states away.
    // volatile: never deleted, duplicated or moved across other volatile asm (the s_nop pair that ends the first K tile);
    // tests/test_isa_budget_cpu.py checks in the emitted code that no other AGPR reader sits between them.
    static ESMK_DEV void mma16_tied(v8 a, v8 b, f32x4& c) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
    static ESMK_DEV _Float16 from(float x) { return (_Float16)x; }
    static ESMK_DEV float to(_Float16 x) { return (float)x; }
};

template <>
struct Op<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static ESMK_DEV f32x16 mma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static ESMK_DEV f32x16 mma_keep_c(v8 a, v8 b, const f32x16& c) {
        f32x16 d;
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    static ESMK_DEV f32x4 mma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static ESMK_DEV void mma16_tied(v8 a, v8 b, f32x4& c) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
    static ESMK_DEV __bf16 from(float x) { return (__bf16)x; }
    static ESMK_DEV float to(__bf16 x) { return (float)x; }
};

// Row of the 32x32 MFMA accumulator held in register r by a lane in half h (= lane >> 5):
// D[row][col = lane & 31], row = (r & 3) + 8 * (r >> 2) + 4 * h.
ESMK_DEV int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Asynchronous 16-byte-per-lane global -> LDS copy.  `lds_wave_base` must be wave uniform;
// lane l lands at lds_wave_base + 16 * l.  The source address is per lane.
ESMK_DEV void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0,
                                     0);
}

ESMK_DEV void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// exact-erf GELU, reference esm/modules.py:17-24:  x * 0.5 * (1 + erf(x / sqrt(2)))
ESMK_DEV float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

// Same function as x * (0.5 + u Q(t)),  u = clamp(x, -4.75, 4.75),  t = 2 u^2 / 4.75^2 - 1,  Q a degree-11 minimax
// fit of erf(u / sqrt 2) / (2u) weighted by u^2 (tools/fit_gelu_poly.py regenerates the coefficients and re-checks the
// bound in emulated fp32): |gelu_fast - gelu_erf| <= 1.4e-6 for |x| <= 4.75 and <= 1.4e-6 |x| beyond, ~100x below the
// fp16 rounding of the value it feeds.  No rcp / exp (quarter rate) and nothing but FMAs, so that two elements share
// one v_pk_fma_f32: ~8 VALU issue cycles per element against ~24 for the erfc(|z|) = poly(t) exp(-z^2) form of
// round 1 (the GELU epilogue of fc1 is VALU bound: 10 k of the tile's 71 k cycles).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kGeluClamp = 4.75f, kGeluK2 = 8.864265680e-02f;  // 2 / 4.75^2
// Horner coefficients of Q, highest power of t first
#define ESMK_GELU_COEF                                                                                       \
    {-5.631324602e-04f, 1.756936894e-03f, -2.548059914e-03f, 4.025654402e-03f, -8.102229796e-03f, 1.411156729e-02f, \
     -2.135194838e-02f, 3.020246327e-02f, -4.060446471e-02f, 5.325455219e-02f, -7.366643846e-02f, 1.487480104e-01f}
// The same form for OPERAND-DTYPE outputs (fc1 -> the A operand of fc2, rounded to 11 / 8 mantissa bits in the same epilogue):
// degree 8, clamp 4 (t = u^2 / 8 - 1).  |gelu_fast<true> - gelu_erf| <= 7.6e-6 ABSOLUTE for |x| <= 4.  Relative to the value
// that is 2.4e-5 where |gelu| > 0.25 (10 x below fp16's half ulp of 1.2 - 2.4e-4), 5.1e-5 above 0.1, and 2.2e-4 — the size of
// the half ulp itself — at |gelu| ~ 0.03 in the negative lobe (x ~ -2.2): the small values are rounded about sqrt 2 worse than
// fp16 alone would; they carry ~1 % of fc2's sum.  Beyond the clamp Phi(4) = 1 - 3.2e-5 stands in for 1: x (1 - 3.2e-5) on
// the right, and a ONE-SIGNED residue -3.2e-5 |x| on the left where the exact value decays to 0 (-1.5e-4 at x = -5, -3.2e-4
// at -10; tests/test_host_cpu.py pins both).  3 of the 17 packed instructions per element pair gone against degree 11
// (round 5; the fc1 epilogue is VALU bound with one wave per SIMD).  fp32
// outputs (the LM head's dense layer) keep the degree-11 set above.  tools/fit_gelu_poly.py --degree 8 --clamp 4.
constexpr float kGeluClampT = 4.0f, kGeluK2T = 1.25e-01f;  // 2 / 4^2
#define ESMK_GELU_COEF_T                                                                                       \
    {1.130852732e-03f, -3.676421475e-03f, 6.586526521e-03f, -1.198850013e-02f, 2.230054513e-02f, -3.696216643e-02f, \
     5.596988276e-02f, -8.431715518e-02f, 1.759489626e-01f}
// T16 = the operand-dtype set
template <bool T16>
struct GeluSet {
    static constexpr int N = T16 ? 9 : 12;
    static constexpr float clamp = T16 ? kGeluClampT : kGeluClamp;
    static constexpr float k2 = T16 ? kGeluK2T : kGeluK2;
    static ESMK_DEV constexpr float coef(int j) {
        constexpr float hi[12] = ESMK_GELU_COEF;
        constexpr float lo[9] = ESMK_GELU_COEF_T;
        return T16 ? lo[j < 9 ? j : 8] : hi[j];
    }
};
template <bool T16 = false>
ESMK_DEV float gelu_fast(float x) {
    using G = GeluSet<T16>;
    const float u = __builtin_amdgcn_fmed3f(x, -G::clamp, G::clamp);
    const float t = __builtin_fmaf(u * G::k2, u, -1.0f);
    float q = G::coef(0);
#pragma unroll
    for (int k = 1; k < G::N; ++k) q = __builtin_fmaf(q, t, G::coef(k));
    return x * __builtin_fmaf(u, q, 0.5f);
}
// In place on four consecutive values (float[4] or a 4-vector), two elements per instruction; bit-identical to
// gelu_fast on each element (same IEEE operations in the same order).  The two Horner chains are interleaved by
// hand: dependent packed FMAs need a wait state that the other chain fills.
template <bool T16 = false, typename V>
ESMK_DEV void gelu_fast_x4(V& v) {
    using G = GeluSet<T16>;
    const f32x2 xa = {v[0], v[1]}, xb = {v[2], v[3]};
    f32x2 ua, ub;
    ua.x = __builtin_amdgcn_fmed3f(xa.x, -G::clamp, G::clamp), ua.y = __builtin_amdgcn_fmed3f(xa.y, -G::clamp, G::clamp);
    ub.x = __builtin_amdgcn_fmed3f(xb.x, -G::clamp, G::clamp), ub.y = __builtin_amdgcn_fmed3f(xb.y, -G::clamp, G::clamp);
    const f32x2 ta = __builtin_elementwise_fma(ua * G::k2, ua, (f32x2)(-1.0f));
    const f32x2 tb = __builtin_elementwise_fma(ub * G::k2, ub, (f32x2)(-1.0f));
    f32x2 qa = (f32x2)(G::coef(0)), qb = (f32x2)(G::coef(0));
#pragma unroll
    for (int k = 1; k < G::N; ++k) {
        qa = __builtin_elementwise_fma(qa, ta, (f32x2)(G::coef(k)));
        qb = __builtin_elementwise_fma(qb, tb, (f32x2)(G::coef(k)));
    }
    const f32x2 ra = xa * __builtin_elementwise_fma(ua, qa, (f32x2)(0.5f));
    const f32x2 rb = xb * __builtin_elementwise_fma(ub, qb, (f32x2)(0.5f));
    v[0] = ra.x, v[1] = ra.y, v[2] = rb.x, v[3] = rb.y;
}

// Eight values at once: FOUR interleaved Horner chains.  A dependent v_pk_fma_f32 needs ~8 cycles before its result can
// be consumed and issues in 4, so two chains keep ONE wave's VALU half busy; kernels with two waves per SIMD fill the gaps
// from the other wave, gemm9 (one wave per SIMD) needs the four chains in its own stream (its GELU epilogue: 15.0k ->
// cycles per 128 x 128 block).  Bit-identical to gelu_fast per element.
template <bool T16 = false, typename V>
ESMK_DEV void gelu_fast_x8(V& v) {
    using G = GeluSet<T16>;
    f32x2 x[4], u[4], t[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = f32x2{v[2 * k], v[2 * k + 1]};
        u[k].x = __builtin_amdgcn_fmed3f(x[k].x, -G::clamp, G::clamp);
        u[k].y = __builtin_amdgcn_fmed3f(x[k].y, -G::clamp, G::clamp);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = __builtin_elementwise_fma(u[k] * G::k2, u[k], (f32x2)(-1.0f));
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (f32x2)(G::coef(0));
#pragma unroll
    for (int j = 1; j < G::N; ++j) {
        __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise serialises the chains again to save registers)
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = __builtin_elementwise_fma(q[k], t[k], (f32x2)(G::coef(j)));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 r = x[k] * __builtin_elementwise_fma(u[k], q[k], (f32x2)(0.5f));
        v[2 * k] = r.x, v[2 * k + 1] = r.y;
    }
}

ESMK_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware bijective remap of a linear workgroup id: workgroup b runs on XCD b % 8, so give
// each XCD a contiguous range of tile ids (neighbouring tiles then share one L2).
ESMK_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
