// attention128.hip — multi-head self-attention core for head_dim 128 (esm2_t48_15B: 40 heads x 128).
//
// Same algorithm and data flow as attention.hip (reference esm/multihead_attention.py:357-394:
// bmm(q,k^T) -> key padding fill -> fp32 softmax -> bmm(probs,v) -> head merge, as one flash-style
// kernel), with the tile geometry of a 128-wide head:
//   q, k      [B,H,T,128]   (scaled / rotated by the QKV epilogue; the 128 dims of a head are stored in
//                            the epilogue's slice order [0..31, 64..95, 32..63, 96..127] — the same order
//                            for q and k, so q.k is unaffected)
//   vt        [B,H,128,Tp]  (V transposed, keys permuted inside groups of 16 as in attention.hip)
//   K tile    64 keys x 256 B in LDS, 16-byte chunk index XOR-swizzled with (row & 15): a ds_read_b128
//             lane group (16 lanes, 16 different rows, one chunk index) touches 16 distinct bank slots
//   V^T tile  128 rows (dv) x 128 B, chunk index XOR-swizzled with (row >> 1) & 7 as in attention.hip
//   one workgroup = 4 waves x 32 query rows; S^T = K.Q^T costs 8 MFMAs per 32 keys, O^T += V^T.P^T keeps four
//   32x32 accumulators per wave.  LDS: 2 stages x (16 + 16 KiB + key bias) = 64.5 KiB (dynamic).
#include "common.h"
#include "kernels.h"
#include <math.h>

namespace esmk {

namespace {
constexpr int HD = 128;
constexpr int K_TILE = 64 * HD * 2;   // 16 KiB
constexpr int V_TILE = HD * 128;      // 16 KiB
constexpr int STAGE = K_TILE + V_TILE + 256;
}  // namespace

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd128_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
    const float* __restrict__ key_bias, const int* __restrict__ seq_info, T* __restrict__ ctx,
    float* __restrict__ lse, int H, int BH, int nq, int Tlen, int Tp, AttnSegs segs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Op<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lm = lane & 31;
    // XCD-grouped 1-D grid (see attention.hip): ids with equal id % 8 share (batch, head)
    int bh, qblk;
    {
        const int id = blockIdx.x;
        const int bh8 = BH & ~7;
        if (id < bh8 * nq) {
            const int r = id >> 3;
            qblk = r % nq;
            bh = (r / nq) * 8 + (id & 7);
        } else {
            const int r = id - bh8 * nq;
            bh = bh8 + r / nq;
            qblk = r % nq;
        }
    }
    const int b = bh / H, head = bh - b * H;
    // token-packed batches: query block qblk is a work item of one segment (see attention.hip)
    int row0 = 0, Tseg = Tlen, qrel = qblk * 128;
    bool seg_pad = false;
    if (segs.work != nullptr) {
        row0 = __builtin_amdgcn_readfirstlane(segs.work[4 * qblk]);
        Tseg = __builtin_amdgcn_readfirstlane(segs.work[4 * qblk + 1]);
        qrel = __builtin_amdgcn_readfirstlane(segs.work[4 * qblk + 2]);
        seg_pad = segs.npad[__builtin_amdgcn_readfirstlane(segs.work[4 * qblk + 3])] > 0;
    }
    const int q0 = qrel + wave * 32;
    const size_t rbase = (size_t)bh * Tlen + row0;

    int kv_end = Tseg;
    bool use_mask = (Tseg & 63) != 0;
    if (segs.work != nullptr) {
        if (seg_pad) use_mask = true;
        else key_bias = nullptr;
    } else if (key_bias != nullptr) {
        if (seq_info != nullptr) {
            if (seq_info[2 * b] > 0) {
                use_mask = true;
                kv_end = seq_info[2 * b + 1];
            }
        } else {
            use_mask = true;
        }
    }
    const int ntiles = (kv_end + 63) >> 6;

    const T* kb = k + rbase * HD;
    const T* vb = vt + (size_t)bh * HD * Tp + row0;

    // Q fragments: Q[q0 + lm][16 ks + 8 h .. +7], ks = 0..7
    V8 qf[8];
    {
        const int qr = min(q0 + lm, Tseg - 1);
        const T* qp = q + (rbase + qr) * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + 16 * ks);
    }

    // staging: K tile = 64 rows x 16 chunks (4 rounds of 256 lanes), V^T tile = 128 rows x 8 chunks (4 rounds)
    const T* gk[4];
    const T* gv[4];
    int krow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pos = j * 256 + tid;
        const int rk = pos >> 4, sk = pos & 15;
        krow[j] = rk;
        gk[j] = kb + (sk ^ (rk & 15)) * 8;  // + key row * 128 (clamped per tile)
        const int rv = pos >> 3, sv = pos & 7;
        gv[j] = vb + (size_t)rv * Tp + (sv ^ ((rv >> 1) & 7)) * 8;  // + tile key offset
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        const int k0 = kt * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kr = min(k0 + krow[j], Tseg - 1);
            glds16(gk[j] + (size_t)kr * HD, base + (j * 256 + wave * 64) * 16);
            glds16(gv[j] + k0, base + K_TILE + (j * 256 + wave * 64) * 16);
        }
        if (use_mask && tid < 64) {
            const int key = k0 + tid;
            float bv = -INFINITY;
            if (key < Tseg) bv = key_bias ? key_bias[(size_t)b * Tlen + row0 + key] : 0.f;
            reinterpret_cast<float*>(base + K_TILE + V_TILE)[tid] = bv;
        }
    };

    // fragment read offsets
    const int krow_off = lm * 256;           // K rows are 256 B
    const int kswz = lm & 15;
    const int vrow_off = lm * 128;           // V^T rows are 128 B
    const int vswz = (lane >> 1) & 7;
    int vxo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) vxo[c] = ((2 * c + h) ^ vswz) << 4;

    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m2 = -INFINITY;
    float lsum = 0.f;

    if (ntiles > 0) stage(0, 0);
    wait_vmcnt0();
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ntiles) stage(cur ^ 1, kt + 1);
        const char* sk = smem + cur * STAGE;
        const char* sv = sk + K_TILE;
        const float* sb = reinterpret_cast<const float*>(sk + K_TILE + V_TILE);

        // ---- S^T = K . Q^T for 64 keys ---------------------------------------------------------
        f32x16 st[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t2][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const V8 kf = *reinterpret_cast<const V8*>(sk + t2 * (32 * 256) + krow_off + (((2 * ks + h) ^ kswz) << 4));
                st[t2] = Op<T>::mma(kf, qf[ks], st[t2]);
            }
        }
        if (use_mask) {  // multihead_attention.py:368-374
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + t2 * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) st[t2][4 * g + e] += bv[e];
                }
        }
        // ---- online softmax (fp32) ---------------------------------------------------------------
        float mx = st[0][0];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m2, mx);  // scores are log2-domain (q carries log2 e, see attention.hip)
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m2 - m_use);
        m2 = m_new;
        float ps = 0.f;
        V8 pf[4];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = __builtin_amdgcn_exp2f(st[t2][8 * ks + e] - m_use);
                    ps += p;
                    pf[2 * t2 + ks][e] = Op<T>::from(p);
                }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        // ---- O^T += V^T . P^T ----------------------------------------------------------------------
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const V8 vf = *reinterpret_cast<const V8*>(sv + d * 4096 + vrow_off + vxo[kk]);
                o[d] = Op<T>::mma(vf, pf[kk], o[d]);
            }
        wait_vmcnt0();
        __syncthreads();
    }

    // ---- normalise and store ctx[b*T + q][head*128 + dv] through a wave-private 8 KiB LDS slice ----------
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / ltot;
    using V4 = typename Op<T>::v4;
    char* wl = smem + wave * 8192;  // rows of 256 B (128 dv), 16 chunks, swizzle (row & 15)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            V4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = Op<T>::from(o[d][4 * g + e] * inv);
            *reinterpret_cast<V4*>(wl + lm * 256 + (((4 * d + g) ^ (lm & 15)) << 4) + 8 * h) = pk;
        }
    T* dst = ctx + ((size_t)b * Tlen + row0) * ((size_t)H * HD) + head * HD;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pc = it * 64 + lane;
        const int r = pc >> 4, c = pc & 15;
        const V8 v = *reinterpret_cast<const V8*>(wl + r * 256 + ((c ^ (r & 15)) << 4));
        if (q0 + r < Tseg) *reinterpret_cast<V8*>(dst + (size_t)(q0 + r) * ((size_t)H * HD) + c * 8) = v;
    }
    const int qrow = q0 + lm;
    if (lse != nullptr && h == 0 && qrow < Tseg) lse[rbase + qrow] = m2 + log2f(ltot);  // log2 domain
}

template <typename TT>
static hipError_t launch128(const void* q, const void* k, const void* vt, const float* key_bias,
                            const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                            hipStream_t st, AttnSegs segs = AttnSegs(), int n_items = 0) {
    static bool attr_set = false;
    auto kern = attn_fwd128_kernel<TT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nq = segs.work != nullptr ? n_items : (T + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(nq * B * H), dim3(256), 2 * STAGE, st, (const TT*)q, (const TT*)k, (const TT*)vt,
                       key_bias, seq_info, (TT*)ctx, lse, H, B * H, nq, T, Tp, segs);
    return hipGetLastError();
}

// token-packed batch (see launch_attention_packed)
hipError_t launch_attention128_packed(const void* q, const void* k, const void* vt, const float* key_bias, void* ctx,
                                      int H, int rows, int Tp, AttnSegs segs, int n_items, int operand_dtype,
                                      hipStream_t st) {
    if (segs.work == nullptr || segs.npad == nullptr || n_items <= 0 || H <= 0 || rows <= 0 || Tp < rows || (Tp & 63))
        return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_BF16)
        return launch128<__bf16>(q, k, vt, key_bias, nullptr, ctx, nullptr, 1, H, rows, Tp, st, segs, n_items);
    return launch128<_Float16>(q, k, vt, key_bias, nullptr, ctx, nullptr, 1, H, rows, Tp, st, segs, n_items);
}

hipError_t launch_attention128(const void* q, const void* k, const void* vt, const float* key_bias,
                               const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                               int operand_dtype, hipStream_t st) {
    if (B <= 0 || H <= 0 || T <= 0 || Tp < T || (Tp & 63)) return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_BF16) return launch128<__bf16>(q, k, vt, key_bias, seq_info, ctx, lse, B, H, T, Tp, st);
    return launch128<_Float16>(q, k, vt, key_bias, seq_info, ctx, lse, B, H, T, Tp, st);
}

// ---------------------------------------------------------------------------------------------
// attention probabilities for need_head_weights / contacts (multihead_attention.py:396-403)
// ---------------------------------------------------------------------------------------------
template <typename T, typename O = float>
__global__ __launch_bounds__(256) void attn_probs128_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ key_bias,
                                                             O* __restrict__ probs, int H, int Tlen, int layer,
                                                             int Ltot) {
    using V8 = typename Op<T>::v8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, lm = lane & 31;
    const int nqb = (Tlen + 127) >> 7;
    const int bh = blockIdx.x / nqb;
    const int b = bh / H, head = bh - b * H;
    const int q0 = (blockIdx.x - bh * nqb) * 128 + wave * 32;
    if (q0 >= Tlen) return;
    V8 qf[8];
    {
        const int qr = min(q0 + lm, Tlen - 1);
        const T* qp = q + ((size_t)bh * Tlen + qr) * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + 16 * ks);
    }
    float row_lse[16], row_keep[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qc = min(q0 + mfma32_row(r, h), Tlen - 1);
        row_lse[r] = lse[(size_t)bh * Tlen + qc];
        row_keep[r] = (key_bias != nullptr && key_bias[(size_t)b * Tlen + qc] != 0.f) ? 0.f : 1.f;
    }
    O* out = probs + (((size_t)b * Ltot + layer) * H + head) * (size_t)Tlen * Tlen;
    for (int k0 = 0; k0 < Tlen; k0 += 32) {
        const int key = k0 + lm;
        const int kc = min(key, Tlen - 1);
        const T* kp = k + ((size_t)bh * Tlen + kc) * HD + 8 * h;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const V8 kf = *reinterpret_cast<const V8*>(kp + 16 * ks);
            s = Op<T>::mma(qf[ks], kf, s);
        }
        const float kbv = (key_bias != nullptr) ? key_bias[(size_t)b * Tlen + kc] : 0.f;
        if (key < Tlen) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = q0 + mfma32_row(r, h);
                if (qr < Tlen) out[(size_t)qr * Tlen + key] = (O)(__builtin_amdgcn_exp2f(s[r] + kbv - row_lse[r]) * row_keep[r]);
            }
        }
    }
}

hipError_t launch_attention_probs128(const void* q, const void* k, const float* lse, const float* key_bias,
                                     float* probs, int B, int H, int T, int layer, int num_layers_total,
                                     int operand_dtype, hipStream_t st, bool lowp) {
    dim3 grid((unsigned)(((T + 127) / 128) * B * H));
    if (lowp) {  // maps in the operand dtype (ESMK_OUT_ATTN_LOWP)
        if (operand_dtype == ESMK_DT_BF16)
            hipLaunchKernelGGL((attn_probs128_kernel<__bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)q,
                               (const __bf16*)k, lse, key_bias, (__bf16*)probs, H, T, layer, num_layers_total);
        else
            hipLaunchKernelGGL((attn_probs128_kernel<_Float16, _Float16>), grid, dim3(256), 0, st, (const _Float16*)q,
                               (const _Float16*)k, lse, key_bias, (_Float16*)probs, H, T, layer, num_layers_total);
        return hipGetLastError();
    }
    if (operand_dtype == ESMK_DT_BF16)
        hipLaunchKernelGGL((attn_probs128_kernel<__bf16>), grid, dim3(256), 0, st, (const __bf16*)q, (const __bf16*)k,
                           lse, key_bias, probs, H, T, layer, num_layers_total);
    else
        hipLaunchKernelGGL((attn_probs128_kernel<_Float16>), grid, dim3(256), 0, st, (const _Float16*)q,
                           (const _Float16*)k, lse, key_bias, probs, H, T, layer, num_layers_total);
    return hipGetLastError();
}

}  // namespace esmk
