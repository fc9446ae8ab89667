// gemm_epi.h — shared by the GEMM kernels (gemm8.hip, gemm9.hip, gemm.hip): LDS geometry, barrier / wait helpers and the
// fused epilogues of a wave's 128 x 64 block (bias, q scale + RoPE + head-major store, V^T store, GELU, fp32 residual
// read-modify-write; reference esm/multihead_attention.py:256-284,395, esm/rotary_embedding.py:11-20, esm/modules.py:17-24,134-140).
#pragma once
#include "common.h"
#include "kernels.h"

namespace esmk {

constexpr int P_UNIT = 128 * 128;           // 16 KiB: 128 rows x 64 operand elements
constexpr int P_BUF = 4 * P_UNIT;           // one K tile
constexpr int P_EPI = 2 * P_BUF;            // offset of the epilogue slices
constexpr int P_SLICE = 4096;               // per wave
constexpr int P_LDS = P_EPI + 8 * P_SLICE;  // 160 KiB

ESMK_DEV void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
ESMK_DEV void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
ESMK_DEV void wait_vmcnt8() { wait_vmcnt<8>(); }

// Global stores a full (unclipped) epilogue issues per wave: the first K tile after it may leave
// them in flight, see the wait bookkeeping in gemm8_kernel.
template <int EPI>
constexpr int epilogue_stores() {
    return (EPI == EPI_STORE_F32 || EPI == EPI_GELU_F32 || EPI == EPI_RESID_F32) ? 32 : 16;
}

// EPI_RESID_F32 row remap of the MSA column-attention block: the GEMM runs on rows ordered (b,c,r)
// (column attention treats every MSA column as a sequence), the residual stream is ordered (b,r,c).
template <bool GEN>
ESMK_DEV int remap_row(const GemmArgs& p, int m) {
    if constexpr (!GEN) return m;
    if (p.rowmap_R == 0) return m;
    const int rc = p.rowmap_R * p.rowmap_C;
    const int b = m / rc, rem = m - b * rc;
    const int c = rem / p.rowmap_R, r = rem - c * p.rowmap_R;
    return (b * p.rowmap_R + r) * p.rowmap_C + c;
}

template <typename T>
ESMK_DEV typename Op<T>::v4 pack4_(float a, float b, float c, float d) {
    typename Op<T>::v4 v;
    v[0] = Op<T>::from(a);
    v[1] = Op<T>::from(b);
    v[2] = Op<T>::from(c);
    v[3] = Op<T>::from(d);
    return v;
}

// --------------------------------------------------------------------------------------------
// epilogue8m: the wave's 128 (m) x 64 (n) block leaves through a private 4 KiB LDS slice in 32-row pieces, so that
// every global store instruction covers whole 128-byte row segments.  Accumulators of 16 x 16 x 32 MFMAs: acc[nj][mi] =
// 16-column block nj0 + nj of the wave's 64 columns x 16-row block mi of its rows, four registers each:
//   normal orientation (MFMA first operand = weight rows):
//     acc[nj][mi][r]:  m = m_base + 16 mi + (lane & 15);  n = n_base + 16 nj + 4 (lane >> 4) + r
//   EPI_V_T (MFMA first operand = activation rows):
//     acc[nj][mi][r]:  n = n_base + 16 nj + (lane & 15);  m = m_base + 16 mi + 4 (lane >> 4) + r
// --------------------------------------------------------------------------------------------
// The bias is already in the accumulators (gemm8_kernel starts every tile from acc = bias) except for
// EPI_V_T, whose bias varies with the lane instead of the register.  FULL = the wave's block lies
// completely inside [0,M) x [0,N): no store is predicated, so the wave issues exactly
// epilogue_stores<EPI>() store instructions.
// NI = number of 32-row pieces of the wave's block: 4 (128 rows), or 2 in the half-height tile mode (HM).
// NT: the aligned global stores carry the non-temporal policy (large launches: the output is not read again by this
// launch and should not push the operand panels out of the XCD's L2).
// LNF (gemm9, EPI_V_T only): LayerNorm-fold consumer — value = ln_rstd[token] * acc + (bias + bias2)[channel]
// (kernels.h, GemmArgs::ln_rstd; the buffer is padded to whole 256-row tiles, so the 4-token loads never leave it).
// COL0 (EPI_V_T inside gemm9's one-launch q / k / v form): the v block starts at column col0 of the launch's N and bias;
// n_base is counted from there.  (A template switch, not a defaulted argument: the plain instantiations keep their code.)
template <typename T, int EPI, bool FULL, bool NOSTORE = false, bool GEN = false, int NI = 4, int NJT = 4, bool NT = false, int NMI = 8, bool LNF = false,
          bool COL0 = false>
ESMK_DEV void epilogue8m(const GemmArgs& p, f32x4 (&acc)[NJT][NMI], int nj0, int m_base, int n_base, int lane,
                         char* wl, size_t out_off, int zo, int zi, int col0 = 0) {
    using V4 = typename Op<T>::v4;
    using V8 = typename Op<T>::v8;
    const int g4 = lane >> 4, l16 = lane & 15;
    if constexpr (!FULL) {
        if constexpr (COL0) {
            if (n_base >= p.N - col0 || m_base >= p.M) return;
        } else {
            if (n_base >= p.N || m_base >= p.M) return;  // wave uniform
        }
    }

    if constexpr (EPI == EPI_V_T) {
        // vt[b][head][dv][Tp], keys permuted inside groups of 16 (4-groups 1 and 2 swapped)
        const int hd = GEN ? p.head_dim : 64;   // 128: the wave's 64 columns are one half of a head
        const int head = n_base / hd;
        const int dv0 = n_base - head * hd;
        T* vt = reinterpret_cast<T*>(p.vt);
        float bvn[4];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            if constexpr (COL0) bvn[nj] = p.bias[col0 + n_base + 16 * nj + l16];
            else bvn[nj] = p.bias[n_base + 16 * nj + l16];
            if constexpr (LNF)
                if (p.bias2 != nullptr) bvn[nj] += p.bias2[(COL0 ? col0 : 0) + n_base + 16 * nj + l16];
        }
        const bool aligned = FULL || (p.T % 32 == 0);  // a 32-token piece = one aligned run of one sequence
        const bool perm = !GEN || (p.vt_rows == 0);  // ESM-2 attention consumes permuted keys, the MSA context GEMM plain ones
        // row index of vt for sequence `sq`: ESM-2 [B,H,64,Tp]; MSA row attention [B,H,R,64,Tp], sq = (b,r)
        auto vt_row0 = [&](int sq) -> size_t {
            if (!GEN || p.vt_rows == 0) return (size_t)(sq * p.H + head) * hd + dv0;
            const int bm = sq / p.vt_rows, r = sq - bm * p.vt_rows;
            return ((size_t)(bm * p.H + head) * p.vt_rows + r) * 64;
        };
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 rs4[2];
            if constexpr (LNF) {
#pragma unroll
                for (int mi2 = 0; mi2 < 2; ++mi2)
                    rs4[mi2] = *reinterpret_cast<const f32x4*>(p.ln_rstd + m_base + 32 * i + 16 * mi2 + 4 * g4);
            }
            // LDS piece: 64 rows (dv) x 64 B (32 tokens); 16-byte chunk c holds 8 token slots
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int dv = 16 * nj + l16;
                const float bv = bvn[nj];
#pragma unroll
                for (int mi2 = 0; mi2 < 2; ++mi2) {
                    // tokens t = 16 mi2 + 4 g4 + r (r = 0..3) of the piece; permuted position: the 4-group index g4 of
                    // the 16-group with its two bits swapped
                    const int chunk = 2 * mi2 + ((aligned && perm) ? (g4 & 1) : (g4 >> 1));
                    const int half = (aligned && perm) ? (g4 >> 1) : (g4 & 1);
                    const f32x4& a = acc[nj0 + nj][2 * i + mi2];
                    if constexpr (LNF) {
                        const f32x4& rs = rs4[mi2];
                        float x0 = __builtin_fmaf(a[0], rs[0], bv), x1 = __builtin_fmaf(a[1], rs[1], bv);
                        float x2 = __builtin_fmaf(a[2], rs[2], bv), x3 = __builtin_fmaf(a[3], rs[3], bv);
                        // The fp32 results go through an opaque barrier before the conversion: left alone the compiler
                        // turns elements 0 and 3 into v_fma_mixlo_f16 (one rounding, fp32 product straight to fp16) and
                        // keeps 1 and 2 as v_pk_fma_f32 + v_cvt_pk_f16_f32 (two roundings), so the fp16 value of a token
                        // depended on its row index modulo 4 and a padded batch differed from the packed one by an ulp.
                        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                        *reinterpret_cast<V4*>(wl + dv * 64 + ((chunk ^ ((dv >> 1) & 3)) << 4) + 8 * half) =
                            pack4_<T>(x0, x1, x2, x3);
                    } else
                    *reinterpret_cast<V4*>(wl + dv * 64 + ((chunk ^ ((dv >> 1) & 3)) << 4) + 8 * half) =
                        pack4_<T>(a[0] + bv, a[1] + bv, a[2] + bv, a[3] + bv);
                }
            }
            const int mp = m_base + 32 * i;
            if (aligned) {
                if (FULL || mp < p.M) {
                    const int b = mp / p.T, t0 = mp - b * p.T;
                    T* base = vt + vt_row0(b) * p.Tp + t0;
                    V8 v[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int pc = it * 64 + lane;
                        const int r = pc >> 2, c = pc & 3;
                        v[it] = *reinterpret_cast<const V8*>(wl + r * 64 + ((c ^ ((r >> 1) & 3)) << 4));
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int pc = it * 64 + lane;
                        const int r = pc >> 2, c = pc & 3;
                        if constexpr (NT) __builtin_nontemporal_store(v[it], reinterpret_cast<V8*>(base + (size_t)r * p.Tp + c * 8));
                        else *reinterpret_cast<V8*>(base + (size_t)r * p.Tp + c * 8) = v[it];
                    }
                }
            } else {
                // any T: per-element stores, the key permutation applied per token
#pragma unroll 1
                for (int idx = lane; idx < 64 * 32; idx += 64) {
                    const int r = idx >> 5, tl = idx & 31;
                    const int m = mp + tl;
                    if (m >= p.M) continue;
                    const T v = *reinterpret_cast<const T*>(wl + r * 64 + (((tl >> 3) ^ ((r >> 1) & 3)) << 4) +
                                                            2 * (tl & 7));
                    const int b = m / p.T, t = m - b * p.T;
                    const int t16 = t & 15;
                    const int tp = perm ? ((t & ~15) | ((((t16 >> 2) & 1) << 3) | (((t16 >> 3) & 1) << 2) | (t16 & 3))) : t;
                    vt[(vt_row0(b) + r) * (size_t)p.Tp + tp] = v;
                }
            }
        }
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        // q and k projections (N = 2E): the wave's 64 columns are exactly one head (head_dim 64)
        const int which = n_base / p.E;  // 0 q, 1 k (wave uniform)
        // head_dim 128: a head is two 64-column slices; the weights are packed so that slice sl holds dims
        // [32 sl, 32 sl + 32) and their rotary partners 64 further up, i.e. the partner of column c is c + 32
        const int hd = GEN ? p.head_dim : 64;
        const int hrem = n_base - which * p.E;
        const int head = hrem / hd;
        const int sl = (hrem - head * hd) >> 6;
        const int rope_ld = hd >> 1;
        T* qk = reinterpret_cast<T*>(which == 0 ? p.q : p.k);
        const float sc = which == 0 ? p.scaling : 1.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int mi2 = 0; mi2 < 2; ++mi2) {
                const int row = 16 * mi2 + l16;
                const int m = min(m_base + 32 * i + row, p.M - 1);
                const int t = (GEN && p.row_pos != nullptr) ? p.row_pos[m] : m % p.T;
                // MSA row attention zeroes q at padded positions (axial_attention.py:85-88)
                const float keep = (GEN && which == 0 && p.row_keep != nullptr) ? p.row_keep[m] : 1.0f;
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) {
                    const int d0 = 16 * nj + 4 * g4;  // first of 4 consecutive dims in [0,32); partner dims 32 further up
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos + (size_t)t * rope_ld + sl * 32 + d0);
                    const f32x4 s = *reinterpret_cast<const f32x4*>(p.sin + (size_t)t * rope_ld + sl * 32 + d0);
                    float y1[4], y2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // bias, q scaling (mha.py:261), x*cos + rotate_half(x)*sin (rotary_embedding.py:11-20); the
                        // contraction is spelled out: gemm9's epilogue computes the same expression, bit for bit
                        const float a1 = acc[nj0 + nj][2 * i + mi2][e] * (sc * keep);
                        const float a2 = acc[nj0 + nj + 2][2 * i + mi2][e] * (sc * keep);
                        y1[e] = __builtin_fmaf(a1, c[e], -(a2 * s[e]));
                        y2[e] = __builtin_fmaf(a2, c[e], a1 * s[e]);
                    }
                    const int chunk = 2 * nj + (g4 >> 1);
                    *reinterpret_cast<V4*>(wl + row * 128 + ((chunk ^ (row & 7)) << 4) + 8 * (g4 & 1)) =
                        pack4_<T>(y1[0], y1[1], y1[2], y1[3]);
                    *reinterpret_cast<V4*>(wl + row * 128 + (((4 + chunk) ^ (row & 7)) << 4) + 8 * (g4 & 1)) =
                        pack4_<T>(y2[0], y2[1], y2[2], y2[3]);
                }
            }
            V8 v[4];  // all LDS reads first, then the stores
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int r = pc >> 3, cc = pc & 7;
                v[it] = *reinterpret_cast<const V8*>(wl + r * 128 + ((cc ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int r = pc >> 3, cc = pc & 7;
                const int mm = m_base + 32 * i + r;
                if (FULL || mm < p.M) {
                    const int b = mm / p.T, tt = mm - b * p.T;
                    *reinterpret_cast<V8*>(qk + ((size_t)(b * p.H + head) * p.T + tt) * hd + sl * 64 + cc * 8) = v[it];
                }
            }
        }
    } else if constexpr (EPI == EPI_STORE_T || EPI == EPI_GELU_T || EPI == EPI_MSA_CTX) {
        T* out = reinterpret_cast<T*>(reinterpret_cast<char*>(p.out) + out_off);
        const int ldc = (GEN && p.ldc > 0) ? p.ldc : p.N;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) {
                    const int row = 16 * mi2 + l16;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nj0 + nj][2 * i + mi2][e];
                    if constexpr (EPI == EPI_GELU_T) gelu_fast_x4<true>(v);
                    *reinterpret_cast<V4*>(wl + row * 128 + (((2 * nj + (g4 >> 1)) ^ (row & 7)) << 4) + 8 * (g4 & 1)) =
                        pack4_<T>(v[0], v[1], v[2], v[3]);
                }
            V8 v[4];  // all LDS reads first, then the stores
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int r = pc >> 3, cc = pc & 7;
                v[it] = *reinterpret_cast<const V8*>(wl + r * 128 + ((cc ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int r = pc >> 3, cc = pc & 7;
                const int m = m_base + 32 * i + r, n = n_base + cc * 8;
                if constexpr (NOSTORE) {
                    asm volatile("" ::"v"(v[it]));
                } else if constexpr (EPI == EPI_MSA_CTX) {
                    // rows m = query column i, the wave's 64 columns = one MSA row r = n_base / 64:
                    // ctx[((zo*R + r)*C + m)*ldc + zi*64 + (n - n_base)]   (axial_attention.py:111-112)
                    if (FULL || (m < p.M && n < p.N))
                        *reinterpret_cast<V8*>(out + ((size_t)(zo * p.ctx_R + (n_base >> 6)) * p.ctx_C + m) * ldc +
                                               zi * 64 + cc * 8) = v[it];
                } else if (FULL || (m < p.M && n < p.N)) {
                    *reinterpret_cast<V8*>(out + (size_t)m * ldc + n) = v[it];
                }
            }
        }
    } else {
        // fp32 outputs: 8 pieces of 32 rows x 32 columns (128-byte row segments)
        float* out = reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + out_off);
        const int ldc = (GEN && p.ldc > 0) ? p.ldc : p.N;
        f32x4 old[4], nxt[4];
        auto load_old = [&](f32x4 (&dst)[4], int piece) {
            const int i = piece >> 1, j = piece & 1;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int m = remap_row<GEN>(p, min(m_base + 32 * i + (pc >> 3), p.M - 1));
                const int n = min(n_base + 32 * j + (pc & 7) * 4, p.N - 4);
                dst[it] = *reinterpret_cast<const f32x4*>(out + (size_t)m * ldc + n);
            }
        };
        if constexpr (EPI == EPI_RESID_F32) load_old(old, 0);
#pragma unroll
        for (int piece = 0; piece < 2 * NI; ++piece) {
            const int i = piece >> 1, j = piece & 1;
            if constexpr (EPI == EPI_RESID_F32)
                if (piece + 1 < 2 * NI) load_old(nxt, piece + 1);  // residual of the next piece in flight
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mi2 = q >> 1, nj2 = q & 1;
                const int row = 16 * mi2 + l16;
                f32x4 v = acc[nj0 + 2 * j + nj2][2 * i + mi2];
                if constexpr (EPI == EPI_GELU_F32) gelu_fast_x4(v);
                *reinterpret_cast<f32x4*>(wl + row * 128 + (((4 * nj2 + g4) ^ (row & 7)) << 4)) = v;
            }
            f32x4 vv[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int r = pc >> 3, cc = pc & 7;
                vv[it] = *reinterpret_cast<const f32x4*>(wl + r * 128 + ((cc ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int r = pc >> 3, cc = pc & 7;
                f32x4 v = vv[it];
                if constexpr (EPI == EPI_RESID_F32)
                    v = f32x4{old[it][0] + v[0], old[it][1] + v[1], old[it][2] + v[2], old[it][3] + v[3]};
                const int m = m_base + 32 * i + r, n = n_base + 32 * j + cc * 4;
                if (FULL || (m < p.M && n < p.N))
                    *reinterpret_cast<f32x4*>(out + (size_t)(EPI == EPI_RESID_F32 ? remap_row<GEN>(p, m) : m) * ldc + n) = v;
            }
            if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
                for (int it = 0; it < 4; ++it) old[it] = nxt[it];
            }
        }
    }
}

}  // namespace esmk
