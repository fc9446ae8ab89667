// elementwise.hip — the HBM-bound kernels of the ESM-2 forward: token statistics, embedding
// gather with token-dropout rescale, LayerNorm, parameter conversion, RoPE tables and the
// contact-prediction head.  All of them stream fp32 rows with 16-byte per-lane accesses; the
// relevant roof is HBM (8 TB/s spec, ~6.3 TB/s achievable), not MFMA.
#include "common.h"
#include "kernels.h"
#include <math.h>
#include <stdlib.h>

#include <algorithm>

namespace esmk {

// ---------------------------------------------------------------------------------------------
// per-sequence token statistics — reference esm/model/esm2.py:82 (padding_mask), :86-92 (token
// dropout ratio), :108-109 (mask dropped when the batch has no pad), and the key padding mask
// of esm/multihead_attention.py:368-374 as an additive 0 / -inf row.
//   scale[b] = 1 - n_mask/n_nonpad   (the fp32 divisor of esm2.py:92)
//   seq_info[2b] = number of pad tokens, seq_info[2b+1] = 1 + index of the last non-pad token
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seq_stats_kernel(const int64_t* __restrict__ tokens, int T,
                                                         int pad_idx, int mask_idx,
                                                         float* __restrict__ scale,
                                                         float* __restrict__ key_bias,
                                                         int* __restrict__ seq_info,
                                                         float* __restrict__ keep) {
    __shared__ int s_mask[4], s_pad[4], s_last[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t* row = tokens + (size_t)b * T;
    int n_mask = 0, n_pad = 0, last = 0;
    for (int t = tid; t < T; t += 256) {
        const int64_t tok = row[t];
        const bool is_pad = tok == pad_idx;
        n_mask += tok == mask_idx;
        n_pad += is_pad;
        if (!is_pad) last = t + 1;
        key_bias[(size_t)b * T + t] = is_pad ? -INFINITY : 0.f;
        if (keep != nullptr) keep[(size_t)b * T + t] = is_pad ? 0.f : 1.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n_mask += __shfl_xor(n_mask, o, 64);
        n_pad += __shfl_xor(n_pad, o, 64);
        last = max(last, __shfl_xor(last, o, 64));
    }
    if ((tid & 63) == 0) {
        s_mask[tid >> 6] = n_mask;
        s_pad[tid >> 6] = n_pad;
        s_last[tid >> 6] = last;
    }
    __syncthreads();
    if (tid == 0) {
        n_mask = s_mask[0] + s_mask[1] + s_mask[2] + s_mask[3];
        n_pad = s_pad[0] + s_pad[1] + s_pad[2] + s_pad[3];
        last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
        const float ratio = (float)n_mask / (float)(T - n_pad);  // esm2.py:91
        scale[b] = 1.0f - ratio;                                   // esm2.py:92 divisor
        seq_info[2 * b] = n_pad;
        seq_info[2 * b + 1] = last;
    }
}

hipError_t launch_seq_stats(const int64_t* tokens, int B, int T, int pad_idx, int mask_idx,
                            int token_dropout, float* scale, float* key_bias, int* seq_info,
                            hipStream_t st, float* keep) {
    (void)token_dropout;
    hipLaunchKernelGGL(seq_stats_kernel, dim3(B), dim3(256), 0, st, tokens, T, pad_idx, mask_idx,
                       scale, key_bias, seq_info, keep);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// token-packed batches (esmk_forward_packed): the same statistics per SEGMENT of one packed row space.
// Workgroup s < n_seg handles segment s (rows [seg[2s], seg[2s] + seg[2s+1])) and the gap behind it (up to the
// next segment's first row, or `rows` after the last one).  The token-dropout divisor of esm2.py:91-92 is
// written per ROW so the embedding kernel can run with "sequences" of one row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void packed_stats_kernel(const int64_t* __restrict__ tokens,
                                                            const int* __restrict__ seg, int n_seg, int rows,
                                                            int pad_idx, int mask_idx,
                                                            float* __restrict__ scale_row,
                                                            float* __restrict__ key_bias,
                                                            int* __restrict__ row_pos,
                                                            int* __restrict__ seg_npad,
                                                            float* __restrict__ keep) {
    __shared__ int s_mask[4], s_pad[4];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int start = seg[2 * s], len = seg[2 * s + 1];
    const int next = (s + 1 < n_seg) ? seg[2 * s + 2] : rows;
    int n_mask = 0, n_pad = 0;
    for (int t = tid; t < len; t += 256) {
        const int64_t tok = tokens[start + t];
        const bool is_pad = tok == pad_idx;
        n_mask += tok == mask_idx;
        n_pad += is_pad;
        key_bias[start + t] = is_pad ? -INFINITY : 0.f;
        row_pos[start + t] = t;
        if (keep != nullptr) keep[start + t] = is_pad ? 0.f : 1.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n_mask += __shfl_xor(n_mask, o, 64);
        n_pad += __shfl_xor(n_pad, o, 64);
    }
    if ((tid & 63) == 0) {
        s_mask[tid >> 6] = n_mask;
        s_pad[tid >> 6] = n_pad;
    }
    __syncthreads();
    n_mask = s_mask[0] + s_mask[1] + s_mask[2] + s_mask[3];
    n_pad = s_pad[0] + s_pad[1] + s_pad[2] + s_pad[3];
    const float ratio = (float)n_mask / (float)(len - n_pad);  // esm2.py:91
    const float den = 1.0f - ratio;                             // esm2.py:92 divisor
    for (int t = tid; t < len; t += 256) scale_row[start + t] = den;
    for (int m = start + len + tid; m < next; m += 256) {  // gap rows: <pad> tokens nobody attends to
        scale_row[m] = 1.0f;
        key_bias[m] = -INFINITY;
        row_pos[m] = 0;
        if (keep != nullptr) keep[m] = 0.f;
    }
    if (tid == 0) seg_npad[s] = n_pad;
}

hipError_t launch_packed_stats(const int64_t* tokens, const int* seg, int n_seg, int rows, int pad_idx,
                               int mask_idx, float* scale_row, float* key_bias, int* row_pos, int* seg_npad,
                               hipStream_t st, float* keep) {
    if (n_seg <= 0 || rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(packed_stats_kernel, dim3(n_seg), dim3(256), 0, st, tokens, seg, n_seg, rows, pad_idx,
                       mask_idx, scale_row, key_bias, row_pos, seg_npad, keep);
    return hipGetLastError();
}

// Gap rows of a token-packed batch are nobody's queries, so the attention kernel leaves their context rows
// unwritten; they must still be FINITE, because the next layer turns them into keys / values that the last key
// tile of the segment in front of them multiplies by probability 0.  Workgroup s clears the gap behind segment s.
__global__ __launch_bounds__(256) void zero_gap_rows_kernel(char* __restrict__ buf, const int* __restrict__ seg,
                                                             int n_seg, int rows, int row_bytes) {
    const int s = blockIdx.x;
    const int first = seg[2 * s] + seg[2 * s + 1];
    const int next = (s + 1 < n_seg) ? seg[2 * s + 2] : rows;
    const size_t n16 = (size_t)(next - first) * (row_bytes >> 4);
    f32x4* dst = reinterpret_cast<f32x4*>(buf + (size_t)first * row_bytes);
    for (size_t i = threadIdx.x; i < n16; i += 256) dst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

hipError_t launch_zero_gap_rows(void* buf, const int* seg, int n_seg, int rows, size_t row_bytes, hipStream_t st) {
    if (n_seg <= 0 || rows <= 0 || row_bytes % 16 != 0 || row_bytes > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(zero_gap_rows_kernel, dim3(n_seg), dim3(256), 0, st, (char*)buf, seg, n_seg, rows, (int)row_bytes);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// learned positions of ESM-1b / ESM-1v — reference esm/model/esm1.py:133 and
// LearnedPositionalEmbedding.forward (esm/modules.py:240-257):
//   x[b,t,:] += embed_positions[cumsum(nonpad)[t] * nonpad[t] + pad_idx]         (one workgroup per sequence)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_positions_kernel(const int64_t* __restrict__ tokens,
                                                             const float* __restrict__ pos_emb,
                                                             float* __restrict__ x, int T, int E, int pad_idx,
                                                             int npos, const int* __restrict__ seg) {
    extern __shared__ int s_pos[];  // [T] position ids + 4 wave totals
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    size_t first = (size_t)b * T;  // first row of the sequence
    int* s_tot = s_pos + T;        // (T = longest sequence)
    if (seg != nullptr) {          // token-packed batch: segment b of one row space
        first = (size_t)seg[2 * b];
        T = seg[2 * b + 1];
    }
    const int64_t* row = tokens + first;
    int carry = 0;
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        const int np = (t < T && row[t] != pad_idx) ? 1 : 0;
        int v = np;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        if (lane == 63) s_tot[wave] = v;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += s_tot[w];
        if (t < T) s_pos[t] = (v + base) * np + pad_idx;
        carry += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        __syncthreads();
    }
    const int e4 = E >> 2;
    for (int idx = tid; idx < T * e4; idx += 256) {
        const int t = idx / e4, k = idx - t * e4;
        const int ps = min(s_pos[t], npos - 1);
        const f32x4 pe = reinterpret_cast<const f32x4*>(pos_emb + (size_t)ps * E)[k];
        f32x4* dst = reinterpret_cast<f32x4*>(x + (first + t) * E) + k;
        f32x4 v = *dst;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += pe[e];
        *dst = v;
    }
}

hipError_t launch_add_positions(const int64_t* tokens, const float* pos_emb, float* x, int B, int T, int E,
                                int pad_idx, int npos, hipStream_t st, const int* seg) {
    if (E % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_positions_kernel, dim3(B), dim3(256), (size_t)(T + 4) * sizeof(int), st, tokens, pos_emb,
                       x, T, E, pad_idx, npos, seg);
    return hipGetLastError();
}

// rows of x multiplied by keep[row] (esm1.py:138-139 when there is no emb_layer_norm_before to fold it into)
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ x, const float* __restrict__ keep,
                                                          size_t n4, int e4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float kp = keep[i / e4];
    f32x4 v = reinterpret_cast<f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= kp;
    reinterpret_cast<f32x4*>(x)[i] = v;
}

hipError_t launch_scale_rows(float* x, const float* keep, int rows, int E, hipStream_t st) {
    const size_t n4 = (size_t)rows * (E / 4);
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, keep, n4, E / 4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// embedding — esm2.py:84 (gather), :87 (zero <mask> rows), :92 (x * 0.88 / (1 - ratio)),
// :94-95 (zero pad rows).  One thread per 4 consecutive channels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tokens,
                                                     const float* __restrict__ table,
                                                     const float* __restrict__ scale,
                                                     float* __restrict__ x, int T, int E4, int vocab,
                                                     int pad_idx, int mask_idx, int token_dropout,
                                                     size_t total4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const size_t row = idx / E4;
    const int c4 = (int)(idx - row * E4);
    int64_t tok = tokens[row];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tok >= 0 && tok < vocab) v = *reinterpret_cast<const f32x4*>(table + (size_t)tok * E4 * 4 + c4 * 4);
    if (token_dropout) {
        if (tok == mask_idx) {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float den = scale[row / T];
        const float keep = (float)(1.0 - 0.15 * 0.8);  // python: 1 - mask_ratio_train
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] * keep) / den;
    }
    if (tok == pad_idx) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(x + idx * 4) = v;
}

hipError_t launch_embed(const int64_t* tokens, const float* table, const float* scale, float* x,
                        int B, int T, int E, int vocab, int pad_idx, int mask_idx,
                        int token_dropout, hipStream_t st) {
    const size_t total4 = (size_t)B * T * (E / 4);
    const unsigned blocks = (unsigned)((total4 + 255) / 256);
    hipLaunchKernelGGL(embed_kernel, dim3(blocks), dim3(256), 0, st, tokens, table, scale, x, T,
                       E / 4, vocab, pad_idx, mask_idx, token_dropout, total4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm — reference esm/modules.py:68-81 (ESM1bLayerNorm == torch.nn.LayerNorm, eps 1e-5,
// biased variance, affine).  One wave per row, the row is held in registers (NCH float4 per
// lane), two-pass mean / variance in fp32, 16-byte loads and 8/16-byte stores.
// Algorithmic traffic per row: 4E read + 2E (operand dtype) and/or 4E (fp32) written.
// ---------------------------------------------------------------------------------------------
// VAR (tools/microbench.py --only ln sweeps it; the engine uses LN_DEFAULT_VARIANT):
//   bit 0: two rows per wave (both rows' loads in flight before the first reduction)
//   bit 1: non-temporal stores (the normalised rows are consumed by the next kernel from HBM/MALL,
//          not from this CU's cache)
//   bit 2: non-temporal loads
template <typename T, int NCH, int VAR>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         T* __restrict__ y, float* __restrict__ y32,
                                                         int rows, int E, LnExtra ex) {
    constexpr int RPW = (VAR & 1) ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int e4 = E >> 2;
    f32x4 v[RPW][NCH];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, rows - 1);
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * E);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < e4) {
                if constexpr (VAR & 4) v[r][i] = __builtin_nontemporal_load(xr + c);
                else v[r][i] = xr[c];
            } else {
                v[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) t += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
        s[r] = t;
    }
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) mean[r] = wave_sum(s[r]) / (float)E;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < e4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[r][i][e] - mean[r];
                    q += d * d;
                }
            }
        }
        s[r] = q;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(s[r]) / (float)E + 1e-5f);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < e4) {
            const f32x4 g = g4[c], bb = b4[c];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                int row = row0 + r;
                if (row >= rows) continue;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[r][i][e] - mean[r]) * rstd[r] * g[e] + bb[e];
                if (ex.row_keep != nullptr) {  // msa_transformer.py:171-172: padded positions are zeroed
                    const float kp = ex.row_keep[row];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= kp;
                }
                if (ex.map_R > 0) {  // input rows ordered (b,r,c) -> output rows ordered (b,c,r)
                    const int rc = ex.map_R * ex.map_C;
                    const int b = row / rc, rem = row - b * rc;
                    const int rr = rem / ex.map_C, cc = rem - rr * ex.map_C;
                    row = (b * ex.map_C + cc) * ex.map_R + rr;
                }
                if (y && ex.x3) {  // f16x3: hi | hi | lo per 64-column K tile (kernels.h LnExtra::x3)
                    typename Op<T>::v4 pk, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pk[e] = Op<T>::from(o[e]);
                        lo[e] = Op<T>::from(o[e] - Op<T>::to(pk[e]));
                    }
                    const int col = c * 4;
                    T* q = y + (size_t)row * ex.ldy + (size_t)(col >> 6) * 192 + (col & 63);
                    *reinterpret_cast<typename Op<T>::v4*>(q) = pk;
                    *reinterpret_cast<typename Op<T>::v4*>(q + 64) = pk;
                    *reinterpret_cast<typename Op<T>::v4*>(q + 128) = lo;
                } else if (y) {
                    typename Op<T>::v4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = Op<T>::from(o[e]);
                    auto* dst = reinterpret_cast<typename Op<T>::v4*>(y + (size_t)row * (ex.ldy > 0 ? ex.ldy : E) + c * 4);
                    if constexpr (VAR & 2) __builtin_nontemporal_store(pk, dst);
                    else *dst = pk;
                }
                if (y32) {
                    auto* dst = reinterpret_cast<f32x4*>(y32 + (size_t)row * E + c * 4);
                    if constexpr (VAR & 2) __builtin_nontemporal_store(o, dst);
                    else *dst = o;
                }
            }
        }
    }
}

// measured on MI355X, 65536 x 1280 rows (profiles/r1_v3_microbench.log): variant 0 3.65 TB/s,
// 1 5.34, 3 5.57, 7 6.02 TB/s of algorithmic traffic
constexpr int LN_DEFAULT_VARIANT = 7;

template <typename T, int VAR>
static hipError_t ln_dispatch_v(const float* x, const float* g, const float* b, void* y, float* y32,
                                int rows, int E, LnExtra ex, hipStream_t st) {
    constexpr int RPW = (VAR & 1) ? 2 : 1;
    const unsigned blocks = (unsigned)((rows + 4 * RPW - 1) / (4 * RPW));
    T* yt = reinterpret_cast<T*>(y);
#define ESMK_LN(N)                                                                                   \
    hipLaunchKernelGGL((layernorm_kernel<T, N, VAR>), dim3(blocks), dim3(256), 0, st, x, g, b, yt, y32, \
                       rows, E, ex)
    if (E <= 512) ESMK_LN(2);
    else if (E <= 1280) ESMK_LN(5);
    else if (E <= 2560) ESMK_LN(10);
    else if (E <= 5120) ESMK_LN(20);
    else return hipErrorInvalidValue;
#undef ESMK_LN
    return hipGetLastError();
}

template <typename T>
static hipError_t ln_dispatch(const float* x, const float* g, const float* b, void* y, float* y32,
                              int rows, int E, int variant, LnExtra ex, hipStream_t st) {
    switch (variant) {
        case 0: return ln_dispatch_v<T, 0>(x, g, b, y, y32, rows, E, ex, st);
        case 1: return ln_dispatch_v<T, 1>(x, g, b, y, y32, rows, E, ex, st);
        case 2: return ln_dispatch_v<T, 2>(x, g, b, y, y32, rows, E, ex, st);
        case 3: return ln_dispatch_v<T, 3>(x, g, b, y, y32, rows, E, ex, st);
        case 5: return ln_dispatch_v<T, 5>(x, g, b, y, y32, rows, E, ex, st);
        case 6: return ln_dispatch_v<T, 6>(x, g, b, y, y32, rows, E, ex, st);
        case 7: return ln_dispatch_v<T, 7>(x, g, b, y, y32, rows, E, ex, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, void* y,
                            float* y32, int rows, int E, int operand_dtype, hipStream_t st) {
    return launch_layernorm_ex(x, gamma, beta, y, y32, rows, E, operand_dtype, LnExtra(), st);
}

hipError_t launch_layernorm_ex(const float* x, const float* gamma, const float* beta, void* y,
                               float* y32, int rows, int E, int operand_dtype, LnExtra ex,
                               hipStream_t st) {
    if (E % 4 != 0 || rows <= 0) return hipErrorInvalidValue;
    // bits 8..11 of operand_dtype: kernel variant + 1 (micro-benchmarks); 0 = engine default
    static const int env_variant = [] {
        const char* e = getenv("ESMK_LN");
        return e ? atoi(e) : LN_DEFAULT_VARIANT;
    }();
    const int vsel = (operand_dtype >> 8) & 0xf;
    const int variant = vsel ? vsel - 1 : env_variant;
    operand_dtype &= 0xff;
    if (operand_dtype == ESMK_DT_BF16)
        return ln_dispatch<__bf16>(x, gamma, beta, y, y32, rows, E, variant, ex, st);
    return ln_dispatch<_Float16>(x, gamma, beta, y, y32, rows, E, variant, ex, st);
}

// ---------------------------------------------------------------------------------------------
// parameter conversion (load time only)
// ---------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ src, D* __restrict__ dst,
                                                       size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) dst[i] = (D)(float)src[i];
}

template <typename S>
static hipError_t convert_from(const S* src, void* dst, int dst_dtype, size_t n, hipStream_t st) {
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    if (dst_dtype == ESMK_DT_F32)
        hipLaunchKernelGGL((convert_kernel<S, float>), dim3(blocks), dim3(256), 0, st, src, (float*)dst, n);
    else if (dst_dtype == ESMK_DT_F16)
        hipLaunchKernelGGL((convert_kernel<S, _Float16>), dim3(blocks), dim3(256), 0, st, src, (_Float16*)dst, n);
    else if (dst_dtype == ESMK_DT_BF16)
        hipLaunchKernelGGL((convert_kernel<S, __bf16>), dim3(blocks), dim3(256), 0, st, src, (__bf16*)dst, n);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_convert(const void* src, int src_dtype, void* dst, int dst_dtype, size_t n,
                          hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (src_dtype == ESMK_DT_F32) return convert_from((const float*)src, dst, dst_dtype, n, st);
    if (src_dtype == ESMK_DT_F16) return convert_from((const _Float16*)src, dst, dst_dtype, n, st);
    if (src_dtype == ESMK_DT_BF16) return convert_from((const __bf16*)src, dst, dst_dtype, n, st);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// 2-D parameter conversion with head-dim padding (load time only).  Models whose head_dim d is smaller
// than the 64 the attention kernels are built for are packed as if every head had 64 dims: dim i of a head
// goes to slot i (i < d/2) or 32 + (i - d/2), so that the rotary partner of slot c is slot c + 32 exactly as
// for d = 64; the remaining slots are zero rows / columns and contribute nothing to q.k or to out_proj.
//   map mode 0: identity;  1: index x = head * d + i  ->  head * 64 + slot(i)
// ---------------------------------------------------------------------------------------------
ESMK_DEV size_t head_pad_index(size_t x, int d) {
    const size_t head = x / d;
    const int i = (int)(x - head * d);
    if (d == 128) {
        // q/k rows of a 128-wide head in the QKV epilogue's slice order: dims [0,32) | [64,96) | [32,64) | [96,128)
        const int half = i >> 6, j = i & 63;
        return head * 128 + (j >> 5) * 64 + half * 32 + (j & 31);
    }
    return head * 64 + (i < d / 2 ? i : 32 + (i - d / 2));
}

// ---------------------------------------------------------------------------------------------
// LayerNorm fold (reference esm/modules.py:120-140: LayerNorm -> q/k/v projections, LayerNorm -> fc1).
// The standalone LayerNorm pass between a residual GEMM and the GEMM that consumes the normalised rows is gone: the
// residual epilogue (gemm9.hip, LNF producer) writes the new rows in the operand dtype together with per-row partial
// sums, ln_finalize_kernel turns those into (mean, rstd), and the consuming GEMM runs on gamma-folded, row-centred
// weights with rstd applied in its epilogue.  rowstats_kernel is the entry of the chain (layer 0, after the embedding).
// ---------------------------------------------------------------------------------------------
// x fp32 [rows, E] -> y[row][c] = T(x - mean) (row stride ldy), mean[row], rstd[row].  Same structure and traffic as
// layernorm_kernel (one wave per two rows, non-temporal accesses); two-pass variance.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, T* __restrict__ y, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out, int rows, int E, int ldy) {
    constexpr int RPW = 2;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int e4 = E >> 2;
    f32x4 v[RPW][NCH];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, rows - 1);
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * E);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            v[r][i] = c < e4 ? __builtin_nontemporal_load(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) t += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
        const float mean = wave_sum(t) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < e4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[r][i][e] -= mean;
                    q += v[r][i][e] * v[r][i][e];
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + 1e-5f);
        const int row = row0 + r;
        if (row >= rows) continue;
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < e4) {
                typename Op<T>::v4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = Op<T>::from(v[r][i][e]);
                __builtin_nontemporal_store(pk, reinterpret_cast<typename Op<T>::v4*>(y + (size_t)row * ldy + c * 4));
            }
        }
    }
}

hipError_t launch_rowstats(const float* x, void* y, float* mean, float* rstd, int rows, int E, int ldy, int operand_dtype,
                           hipStream_t st) {
    if (E % 4 != 0 || rows <= 0 || ldy < E) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((rows + 7) / 8);
#define ESMK_RS(TT, N) hipLaunchKernelGGL((rowstats_kernel<TT, N>), dim3(blocks), dim3(256), 0, st, x, (TT*)y, mean, rstd, rows, E, ldy)
#define ESMK_RS_T(TT)                  \
    if (E <= 512) ESMK_RS(TT, 2);      \
    else if (E <= 1280) ESMK_RS(TT, 5);  \
    else if (E <= 2560) ESMK_RS(TT, 10); \
    else if (E <= 5120) ESMK_RS(TT, 20); \
    else return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_BF16) { ESMK_RS_T(__bf16) } else { ESMK_RS_T(_Float16) }
#undef ESMK_RS_T
#undef ESMK_RS
    return hipGetLastError();
}

// part[row][parts][2] = (sum d, sum d^2) over 128-column slabs, d = x_new - mean_prev[row]  ->  mean[row] (updated in
// place: the next producer subtracts it), rstd[row].  Summed in slab order: deterministic.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                                           float* __restrict__ rstd, int rows, int parts, float inv_e) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const f32x2* pr = reinterpret_cast<const f32x2*>(part) + (size_t)row * parts;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < parts; ++k) {
        const f32x2 v = pr[k];
        s1 += v.x;
        s2 += v.y;
    }
    const float dm = s1 * inv_e;                       // mean of d: small against the spread by construction
    const float var = fmaxf(s2 * inv_e - dm * dm, 0.f);
    mean[row] += dm;
    rstd[row] = 1.0f / sqrtf(var + 1e-5f);
}

hipError_t launch_ln_finalize(const float* part, float* mean, float* rstd, int rows, int parts, int E, hipStream_t st) {
    if (rows <= 0 || parts <= 0 || E <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, part, mean, rstd, rows, parts,
                       1.0f / (float)E);
    return hipGetLastError();
}

// Load time: one weight row n of a linear layer that follows a LayerNorm (q/k/v projections, fc1) -> its folded image
//   dst[map(n)][k] = T(w[n][k] gamma[k] - mean_k(w[n][.] gamma[.]))      (k < cols; pad columns of the image stay zero)
//   bias2[map(n)]  = sum_k w[n][k] beta[k]
// so that  LayerNorm(x) . w[n]^T + b[n]  =  rstd * ((x - c) . dst[n]^T) + b[n] + bias2[n]  for ANY per-row constant c
// (the centred rows sum to zero up to the rounding of the image).  One workgroup per row, fp32 sums in a fixed order.
template <typename S, typename D>
__global__ __launch_bounds__(256) void fold_weight_kernel(const S* __restrict__ src, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, D* __restrict__ dst,
                                                           float* __restrict__ bias2, int cols, size_t dst_ld, int row_map, int d) {
    __shared__ float red[2][4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const S* w = src + (size_t)n * cols;
    float sg = 0.f, sb = 0.f;
    for (int k = tid; k < cols; k += 256) {
        const float v = (float)w[k];
        sg += v * gamma[k];
        sb += v * beta[k];
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if ((tid & 63) == 0) red[0][tid >> 6] = sg, red[1][tid >> 6] = sb;
    __syncthreads();
    const float rowmean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)cols;
    const float wb = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const size_t nn = row_map ? head_pad_index((size_t)n, d) : (size_t)n;
    for (int k = tid; k < cols; k += 256) dst[nn * dst_ld + k] = (D)((float)w[k] * gamma[k] - rowmean);
    if (tid == 0) bias2[nn] = wb;
}

hipError_t launch_fold_weight(const void* src, int src_dtype, const float* gamma, const float* beta, void* dst, int dst_dtype,
                              float* bias2, size_t rows, size_t cols, size_t dst_ld, int row_map, int d, hipStream_t st) {
    if (rows == 0 || cols == 0) return hipSuccess;
#define ESMK_FW(ST, DT) \
    hipLaunchKernelGGL((fold_weight_kernel<ST, DT>), dim3((unsigned)rows), dim3(256), 0, st, (const ST*)src, gamma, beta, (DT*)dst, bias2, \
                       (int)cols, dst_ld, row_map, d)
#define ESMK_FW_S(ST)                                     \
    if (dst_dtype == ESMK_DT_F16) ESMK_FW(ST, _Float16);  \
    else if (dst_dtype == ESMK_DT_BF16) ESMK_FW(ST, __bf16); \
    else return hipErrorInvalidValue;
    if (src_dtype == ESMK_DT_F32) { ESMK_FW_S(float) }
    else if (src_dtype == ESMK_DT_F16) { ESMK_FW_S(_Float16) }
    else if (src_dtype == ESMK_DT_BF16) { ESMK_FW_S(__bf16) }
    else return hipErrorInvalidValue;
#undef ESMK_FW_S
#undef ESMK_FW
    return hipGetLastError();
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void convert2d_kernel(const S* __restrict__ src, D* __restrict__ dst,
                                                         size_t rows, size_t cols, size_t dst_ld, int row_map,
                                                         int col_map, int d) {
    const size_t n = rows * cols;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const size_t r = i / cols, c = i - r * cols;
        const size_t rr = row_map ? head_pad_index(r, d) : r;
        const size_t cc = col_map ? head_pad_index(c, d) : c;
        dst[rr * dst_ld + cc] = (D)(float)src[i];
    }
}

template <typename S>
static hipError_t convert2d_from(const S* src, void* dst, int dst_dtype, size_t rows, size_t cols, size_t dst_ld,
                                 int row_map, int col_map, int d, hipStream_t st) {
    const unsigned blocks = (unsigned)std::min<size_t>((rows * cols + 255) / 256, 4096);
#define ESMK_C2D(DT) \
    hipLaunchKernelGGL((convert2d_kernel<S, DT>), dim3(blocks), dim3(256), 0, st, src, (DT*)dst, rows, cols, dst_ld, \
                       row_map, col_map, d)
    if (dst_dtype == ESMK_DT_F32) ESMK_C2D(float);
    else if (dst_dtype == ESMK_DT_F16) ESMK_C2D(_Float16);
    else if (dst_dtype == ESMK_DT_BF16) ESMK_C2D(__bf16);
    else return hipErrorInvalidValue;
#undef ESMK_C2D
    return hipGetLastError();
}

// THIRD (precision mode f16x3, esmk_config::weight_split 4): every 64-column K tile as hi | lo | hi (rows of 3 dst_ld) — against
// activation rows laid out hi | hi | lo (layernorm_kernel LnExtra::x3, gemm9's X3O GELU epilogue, attn_fwd_kernel X3) a PLAIN GEMM over K' = 3 K computes
// A_hi W_hi + A_hi W_lo + A_lo W_hi, i.e. the product of the operands to ~20 bits each (A_lo W_lo, 2^-22, is dropped).
template <typename S, bool THIRD = false>
__global__ __launch_bounds__(256) void convert2d_split_kernel(const S* __restrict__ src, _Float16* __restrict__ dst,
                                                               size_t rows, size_t cols, size_t dst_ld, int row_map,
                                                               int col_map, int d) {
    const size_t n = rows * cols;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const size_t r = i / cols, c = i - r * cols;
        const size_t rr = row_map ? head_pad_index(r, d) : r;
        const size_t cc = col_map ? head_pad_index(c, d) : c;
        const float w = (float)src[i];
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        _Float16* q = dst + rr * (THIRD ? 3 : 2) * dst_ld + (cc >> 6) * (THIRD ? 192 : 128) + (cc & 63);
        q[0] = hi;
        q[64] = lo;
        if constexpr (THIRD) q[128] = hi;
    }
}

hipError_t launch_convert2d_split(const void* src, int src_dtype, void* dst, size_t rows, size_t cols, size_t dst_ld,
                                  int row_map, int col_map, int d, hipStream_t st, int parts) {
    if (rows * cols == 0) return hipSuccess;
    if (parts != 2 && parts != 3) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)std::min<size_t>((rows * cols + 255) / 256, 4096);
#define ESMK_C2S(ST)                                                                                                          \
    if (parts == 3)                                                                                                           \
        hipLaunchKernelGGL((convert2d_split_kernel<ST, true>), dim3(blocks), dim3(256), 0, st, (const ST*)src, (_Float16*)dst, rows, \
                           cols, dst_ld, row_map, col_map, d);                                                                \
    else                                                                                                                      \
        hipLaunchKernelGGL((convert2d_split_kernel<ST, false>), dim3(blocks), dim3(256), 0, st, (const ST*)src, (_Float16*)dst, rows, \
                           cols, dst_ld, row_map, col_map, d)
    if (src_dtype == ESMK_DT_F32) { ESMK_C2S(float); }
    else if (src_dtype == ESMK_DT_F16) { ESMK_C2S(_Float16); }
    else if (src_dtype == ESMK_DT_BF16) { ESMK_C2S(__bf16); }
    else return hipErrorInvalidValue;
#undef ESMK_C2S
    return hipGetLastError();
}

hipError_t launch_convert2d(const void* src, int src_dtype, void* dst, int dst_dtype, size_t rows, size_t cols,
                            size_t dst_ld, int row_map, int col_map, int d, hipStream_t st) {
    if (rows * cols == 0) return hipSuccess;
    if (src_dtype == ESMK_DT_F32) return convert2d_from((const float*)src, dst, dst_dtype, rows, cols, dst_ld, row_map, col_map, d, st);
    if (src_dtype == ESMK_DT_F16) return convert2d_from((const _Float16*)src, dst, dst_dtype, rows, cols, dst_ld, row_map, col_map, d, st);
    if (src_dtype == ESMK_DT_BF16) return convert2d_from((const __bf16*)src, dst, dst_dtype, rows, cols, dst_ld, row_map, col_map, d, st);
    return hipErrorInvalidValue;
}

__global__ __launch_bounds__(256) void copy_f32_kernel(const float* __restrict__ src,
                                                        float* __restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride)
        reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
}

hipError_t launch_copy_f32(const float* src, float* dst, size_t n, hipStream_t st) {
    if (n % 4 != 0) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(copy_f32_kernel, dim3(blocks), dim3(256), 0, st, src, dst, n4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Mean over the residue rows of every sequence — reference scripts/extract.py:113-116
// (`t[i, 1 : truncate_len + 1].mean(0)`): out[b, e] = mean of x[b, first .. first + cnt_b), cnt_b =
// min(count[b], T - first); an empty slice gives NaN as torch.mean does.  Workgroup = (256-column slab,
// sequence): the four waves each sum a quarter of the rows (16 B per lane, whole 1-KiB / 512-B row segments),
// then combine through LDS in wave order — deterministic, no atomics.  HBM bound: reads the tensor once.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void masked_row_mean_kernel(const T* __restrict__ x, const int* __restrict__ count,
                                                              float* __restrict__ out, int T_rows, int E, int first) {
    __shared__ f32x4 part[4][64];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = blockIdx.x * 256 + lane * 4;
    int cnt = count[b];
    cnt = cnt < 0 ? 0 : (cnt > T_rows - first ? T_rows - first : cnt);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < E) {
        const T* base = x + ((size_t)b * T_rows + first) * E + col;
        for (int r = wave; r < cnt; r += 4) {
            const T* p = base + (size_t)r * E;
            if constexpr (sizeof(T) == 4) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
                acc += v;
            } else {
                typename Op<T>::v4 v = *reinterpret_cast<const typename Op<T>::v4*>(p);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += (float)v[i];
            }
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col < E) {
        f32x4 s = part[0][lane];
        s += part[1][lane];
        s += part[2][lane];
        s += part[3][lane];
        const float inv = 1.0f / (float)cnt;  // cnt == 0: 0 * inf = NaN, the mean of an empty slice
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = s[i] * inv;
        *reinterpret_cast<f32x4*>(out + (size_t)b * E + col) = o;
    }
}

hipError_t launch_masked_row_mean(const void* x, int x_dtype, const int* count, float* out, int B, int T, int E,
                                  int first, hipStream_t st) {
    if (E % 4 != 0 || B <= 0 || T <= 0 || first < 0 || first > T) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((E + 255) / 256), (unsigned)B);
    if (x_dtype == ESMK_DT_F32)
        hipLaunchKernelGGL(masked_row_mean_kernel<float>, grid, dim3(256), 0, st, (const float*)x, count, out, T, E, first);
    else if (x_dtype == ESMK_DT_F16)
        hipLaunchKernelGGL(masked_row_mean_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)x, count, out, T, E, first);
    else if (x_dtype == ESMK_DT_BF16)
        hipLaunchKernelGGL(masked_row_mean_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, count, out, T, E, first);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// RoPE tables — reference esm/rotary_embedding.py:47-61: t = arange(T) (fp32), freqs = t x
// inv_freq (fp32 product), cos/sin in fp32.  Only the d/2 distinct columns are stored (the
// reference duplicates them with cat(freqs, freqs)).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_table_kernel(const float* __restrict__ inv_freq,
                                                          float* __restrict__ cs,
                                                          float* __restrict__ sn, int T, int half) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * half) return;
    const int t = idx / half, i = idx - t * half;
    const float ang = (float)t * inv_freq[i];
    cs[idx] = cosf(ang);
    sn[idx] = sinf(ang);
}

hipError_t launch_rope_table(const float* inv_freq, float* cos, float* sin, int T, int half,
                             hipStream_t st) {
    const unsigned blocks = (unsigned)((T * half + 255) / 256);
    hipLaunchKernelGGL(rope_table_kernel, dim3(blocks), dim3(256), 0, st, inv_freq, cos, sin, T, half);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// contact head — reference esm/modules.py:338-357 with symmetrize / apc (modules.py:27-41).
//
// With A'_c the eos-masked (modules.py:340-343) and cropped (modules.py:344-347) map of channel c,
// S_c = A'_c + A'_c^T, r_c[i] = sum_j S_c[i][j], t_c = sum_i r_c[i]:
//   logit[i][j] = sum_c w_c S_c[i][j] - sum_c (w_c / t_c) r_c[i] r_c[j] + b ;  out = sigmoid(logit)
// which is apc(symmetrize(A')) fed to the 1-output regression without materialising [C,T,T]
// copies.  Pass 1 (contact_sums_kernel) produces r_c and t_c, pass 2 (contact_out_kernel) the
// logits; the attention tensor is read twice in total.
// ---------------------------------------------------------------------------------------------
constexpr int CS_COLS = 1024;  // columns handled per sweep by one block (16 per lane)

__global__ __launch_bounds__(256) void contact_sums_kernel(const float* __restrict__ attn,
                                                            const int64_t* __restrict__ tokens,
                                                            float* __restrict__ rsum,
                                                            float* __restrict__ tsum, int C, int T,
                                                            int eos_idx, int bos, int eos) {
    extern __shared__ float s_part[];  // [4][S] column partials + [S] row sums
    const int S = T - bos - eos;
    const int bc = blockIdx.x;
    const int b = bc / C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* A = attn + (size_t)bc * T * T;
    const int64_t* tok = tokens + (size_t)b * T;
    float* s_row = s_part + 4 * S;

    for (int j0 = 0; j0 < S; j0 += CS_COLS) {
        float col[16];
        float cm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            col[k] = 0.f;
            const int j = j0 + lane + 64 * k;
            cm[k] = (j < S && !(eos && tok[j + bos] == eos_idx)) ? 1.f : 0.f;
        }
        for (int i = wave; i < S; i += 4) {
            const float rm = (eos && tok[i + bos] == eos_idx) ? 0.f : 1.f;
            const float* row = A + (size_t)(i + bos) * T + bos;
            float rs = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int j = j0 + lane + 64 * k;
                float v = 0.f;
                if (j < S) v = row[j] * rm * cm[k];
                rs += v;
                col[k] += v;
            }
            rs = wave_sum(rs);
            if (lane == 0) {
                if (j0 == 0) s_row[i] = rs;
                else s_row[i] += rs;
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = j0 + lane + 64 * k;
            if (j < S) s_part[wave * S + j] = col[k];
        }
    }
    __syncthreads();
    float tot = 0.f;
    for (int i = threadIdx.x; i < S; i += 256) {
        const float r = s_row[i] + ((s_part[i] + s_part[S + i]) + (s_part[2 * S + i] + s_part[3 * S + i]));
        rsum[(size_t)bc * S + i] = r;
        tot += r;
    }
    tot = wave_sum(tot);
    __syncthreads();
    if (lane == 0) s_part[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) tsum[bc] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ __launch_bounds__(256) void contact_out_kernel(const float* __restrict__ attn,
                                                           const int64_t* __restrict__ tokens,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ rsum,
                                                           const float* __restrict__ tsum,
                                                           float* __restrict__ out, int C, int T,
                                                           int eos_idx, int bos, int eos) {
    __shared__ float s_t[32][33];
    const int S = T - bos - eos;
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t* tok = tokens + (size_t)b * T;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float mi[4], mj, mjt[4], mit;
    // masks for the direct tile (rows i0+ty+8k, col j0+tx) and the mirrored tile
    mj = (j0 + tx < S && !(eos && tok[j0 + tx + bos] == eos_idx)) ? 1.f : 0.f;
    mit = (i0 + tx < S && !(eos && tok[i0 + tx + bos] == eos_idx)) ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + ty + 8 * k, j = j0 + ty + 8 * k;
        mi[k] = (i < S && !(eos && tok[i + bos] == eos_idx)) ? 1.f : 0.f;
        mjt[k] = (j < S && !(eos && tok[j + bos] == eos_idx)) ? 1.f : 0.f;
    }
    for (int c = 0; c < C; ++c) {
        const float* A = attn + ((size_t)b * C + c) * T * T;
        const float wc = w[c];
        const float wt = wc / tsum[(size_t)b * C + c];
        const float* r = rsum + ((size_t)b * C + c) * S;
        // mirrored tile A'[j0+ty+8k][i0+tx] -> LDS (coalesced along i), read back transposed
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + ty + 8 * k, i = i0 + tx;
            float v = 0.f;
            if (j < S && i < S) v = A[(size_t)(j + bos) * T + i + bos] * mjt[k] * mit;
            s_t[ty + 8 * k][tx] = v;
        }
        __syncthreads();
        const float rj = (j0 + tx < S) ? r[j0 + tx] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + ty + 8 * k, j = j0 + tx;
            float v = 0.f;
            if (i < S && j < S) v = A[(size_t)(i + bos) * T + j + bos] * mi[k] * mj;
            const float sym = v + s_t[tx][ty + 8 * k];
            const float ri = (i < S) ? r[i] : 0.f;
            acc[k] += wc * sym - wt * ri * rj;
        }
    }
    const float bb = bias ? bias[0] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + ty + 8 * k, j = j0 + tx;
        if (i < S && j < S) {
            const float z = acc[k] + bb;
            out[((size_t)b * S + i) * S + j] = 1.0f / (1.0f + expf(-z));
        }
    }
}

hipError_t launch_contacts(const float* attn, const int64_t* tokens, const float* w,
                           const float* b, float* scratch, float* out, int B, int C, int T,
                           int eos_idx, int prepend_bos, int append_eos, hipStream_t st) {
    const int bos = prepend_bos ? 1 : 0, eos = append_eos ? 1 : 0;
    const int S = T - bos - eos;
    if (S <= 0) return hipErrorInvalidValue;
    float* rsum = scratch;                    // [B*C*S]
    float* tsum = scratch + (size_t)B * C * S;  // [B*C]
    const size_t lds = (size_t)5 * S * sizeof(float) + 64;
    if (lds > 64 * 1024) return hipErrorInvalidValue;  // S <= ~3270
    hipLaunchKernelGGL(contact_sums_kernel, dim3(B * C), dim3(256), lds, st, attn, tokens, rsum,
                       tsum, C, T, eos_idx, bos, eos);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nt = (S + 31) / 32;
    hipLaunchKernelGGL(contact_out_kernel, dim3(nt, nt, B), dim3(256), 0, st, attn, tokens, w, b,
                       rsum, tsum, out, C, T, eos_idx, bos, eos);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// MSA Transformer embedding — reference esm/model/msa_transformer.py:152-165 and
// LearnedPositionalEmbedding.forward (esm/modules.py:240-257).  One workgroup per MSA row (b,r):
//   x[b,r,c,:] = embed_tokens[tok] + embed_positions[cumsum(nonpad)[c] * nonpad[c] + pad_idx]
//                + msa_position_embedding[r]                                    (fp32)
// plus the padding bookkeeping of the axial layers:
//   keep[b,r,c] = 1 - pad (msa_transformer.py:171-172, axial_attention.py:85-88),
//   col_fill[(b,c), r] = pad (column attention key mask, axial_attention.py:211-215; (b,c)-major),
//   any_pad[0] |= pad (the reference drops the mask when the batch has no pad, msa_transformer.py:153-155).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msa_embed_kernel(const int64_t* __restrict__ tokens,
                                                         const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb,
                                                         const float* __restrict__ msa_pos,
                                                         float* __restrict__ x, float* __restrict__ keep,
                                                         float* __restrict__ col_fill,
                                                         int* __restrict__ any_pad, int R, int C, int D,
                                                         int vocab, int pad_idx, int npos) {
    // grid (B*R, column chunks): one workgroup per MSA row left 128 of the 256 CUs idle and serialised 1.5 MB of
    // stores per workgroup (346 us for a 128 x 513 MSA); every chunk recomputes the row's (cheap) position scan
    // and writes only its own columns.
    extern __shared__ int s_pos[];  // [C] position ids, then 4 wave totals
    const int br = blockIdx.x;      // b * R + r
    const int cper = (C + gridDim.y - 1) / gridDim.y;
    const int c_lo = blockIdx.y * cper, c_hi = min(C, c_lo + cper);
    const int b = br / R, r = br - b * R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t* row = tokens + (size_t)br * C;
    int* s_tot = s_pos + C;
    // inclusive prefix sum of the non-pad flags over the row, 256 columns per sweep
    int carry = 0;
    bool saw_pad = false;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + tid;
        const int np = (c < C && row[c] != pad_idx) ? 1 : 0;
        saw_pad |= (c < C && np == 0);
        int v = np;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if (lane >= o) v += t;
        }
        if (lane == 63) s_tot[wave] = v;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += s_tot[w];
        if (c < C) {
            s_pos[c] = (v + base) * np + pad_idx;
            if (c >= c_lo && c < c_hi) {
                keep[(size_t)br * C + c] = (float)np;
                col_fill[((size_t)b * C + c) * R + r] = np ? 0.f : 1.f;
            }
        }
        carry += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        __syncthreads();
    }
    if (saw_pad && blockIdx.y == 0) atomicOr(any_pad, 1);
    const int d4 = D >> 2;
    const f32x4* mp = msa_pos ? reinterpret_cast<const f32x4*>(msa_pos + (size_t)r * D) : nullptr;
    for (int idx = tid; idx < (c_hi - c_lo) * d4; idx += 256) {
        const int c = c_lo + idx / d4, k = idx % d4;
        const int64_t tok = row[c];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tok >= 0 && tok < vocab) v = reinterpret_cast<const f32x4*>(tok_emb + (size_t)tok * D)[k];
        const int ps = min(s_pos[c], npos - 1);
        const f32x4 pe = reinterpret_cast<const f32x4*>(pos_emb + (size_t)ps * D)[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += pe[e];
        if (mp) {
            const f32x4 m4 = mp[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += m4[e];
        }
        reinterpret_cast<f32x4*>(x + ((size_t)br * C + c) * D)[k] = v;
    }
}

hipError_t launch_msa_embed(const int64_t* tokens, const float* tok_emb, const float* pos_emb,
                            const float* msa_pos, float* x, float* keep, float* col_fill, int* any_pad,
                            int B, int R, int C, int D, int vocab, int pad_idx, int npos, hipStream_t st) {
    if (D % 4 != 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(any_pad, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    const unsigned chunks = (unsigned)std::max(1, std::min(16, (C + 63) / 64));
    hipLaunchKernelGGL(msa_embed_kernel, dim3(B * R, chunks), dim3(256), (size_t)(C + 4) * sizeof(int), st, tokens,
                       tok_emb, pos_emb, msa_pos, x, keep, col_fill, any_pad, R, C, D, vocab, pad_idx, npos);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Row-attention softmax — reference esm/axial_attention.py:96-100 (columns that are padded in row 0
// are filled with -10000) and :127 (softmax over the key column j).  One wave per (b,h,i) row.
//   scores fp32 [B*H, C, ldp]  ->  probs (operand dtype) [B*H, C, ldp], columns >= C zeroed (they are the
//   K padding of the context GEMM), and optionally fp32 row_attentions[b, layer, h, i, :] (msa_transformer.py:195-196)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void msa_row_softmax_kernel(const float* __restrict__ scores,
                                                               const float* __restrict__ keep,
                                                               const int* __restrict__ any_pad,
                                                               T* __restrict__ probs,
                                                               float* __restrict__ attn_out, int B, int H,
                                                               int R, int C, int ldp, int layer, int Ltot,
                                                               int nslice) {
    const int lane = threadIdx.x & 63;
    const int rowid = blockIdx.x * 4 + (threadIdx.x >> 6);  // (b*H + h)*C + i
    const int total = B * H * C;
    if (rowid >= total) return;
    const int bh = rowid / C, i = rowid - bh * C;
    const int b = bh / H, h = bh - b * H;
    const bool masked = any_pad[0] != 0;
    // scores [B, nslice, H, C, ldp]: the tied contraction over the R rows arrives as nslice partial maps
    const size_t slice_stride = (size_t)H * C * ldp;
    const float* srow = scores + ((size_t)b * nslice * H + h) * C * ldp + (size_t)i * ldp;
    const float* k0 = keep + (size_t)b * R * C;  // row 0 of MSA b: keep[b,0,j]
    constexpr int MAXJ = 16;                       // columns per lane: C <= 1024
    float v[MAXJ];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
        const int j = lane + 64 * q;
        float s = -INFINITY;
        if (j < C) {
            s = srow[j];
            for (int sl = 1; sl < nslice; ++sl) s += srow[sl * slice_stride + j];  // fixed order: deterministic
            if (masked && k0[j] == 0.f) s = -10000.f;
        }
        v[q] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
        v[q] = (lane + 64 * q < C) ? expf(v[q] - mx) : 0.f;
        sum += v[q];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* prow = probs + (size_t)rowid * ldp;
    float* arow = attn_out ? attn_out + ((((size_t)b * Ltot + layer) * H + h) * C + i) * C : nullptr;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
        const int j = lane + 64 * q;
        if (j < ldp) prow[j] = Op<T>::from(v[q] * inv);
        if (arow && j < C) arow[j] = v[q] * inv;
    }
}

hipError_t launch_msa_row_softmax(const float* scores, const float* keep, const int* any_pad, void* probs,
                                  float* attn_out, int B, int H, int R, int C, int ldp, int layer,
                                  int num_layers_total, int operand_dtype, hipStream_t st, int nslice) {
    if (C > 1024 || ldp > 1024 || ldp < C || nslice < 1) return hipErrorInvalidValue;
    dim3 grid((unsigned)((B * H * C + 3) / 4));
    if (operand_dtype == ESMK_DT_BF16)
        hipLaunchKernelGGL((msa_row_softmax_kernel<__bf16>), grid, dim3(256), 0, st, scores, keep, any_pad,
                           (__bf16*)probs, attn_out, B, H, R, C, ldp, layer, num_layers_total, nslice);
    else
        hipLaunchKernelGGL((msa_row_softmax_kernel<_Float16>), grid, dim3(256), 0, st, scores, keep, any_pad,
                           (_Float16*)probs, attn_out, B, H, R, C, ldp, layer, num_layers_total, nslice);
    return hipGetLastError();
}

}  // namespace esmk
