// common.h — shared device helpers for the gfx950 (MI355X / CDNA4) ESM-2 engine.
// Written for gfx950 only: 64-lane wavefronts, v_mfma_f32_32x32x16_{f16,bf16},
// global_load_lds_dwordx4, 160 KiB LDS per CU.
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

// Same function through erfc(|z|) ~= poly(t) exp(-z^2), t = 1/(1 + p|z|) (Abramowitz-Stegun 7.1.26,
// |error| <= 1.5e-7): |gelu_fast - gelu_erf| <= 3.4e-7 over the whole real line (checked against
// float64 on 2e6 points), i.e. ~1000x below the fp16 rounding of the value it feeds; 1 rcp + 1 exp2
// + 9 FMA/MUL instead of the branchy libm erff.
ESMK_DEV float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly =
        t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float q = poly * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);  // erfc(|z|)
    const float hq = 0.5f * x * q;
    return x >= 0.f ? x - hq : hq;
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
