// contacts.hip — contact maps WITHOUT the [B,L,H,T,T] attention tensor (fused form).
//
// Reference: ESM2.forward collects every layer's head-wise attention weights (esm/model/esm2.py:119-121,
// 132-139; 2.77 GB of fp32 per 1024-token sequence for the 650M model) and hands them to
// ContactPredictionHead.forward (esm/modules.py:338-357): eos mask, crop of the first/last position,
// symmetrize + apc (modules.py:27-41) per channel c = (layer, head), a 1-output regression, sigmoid.
// With m[t] = 1 for residue tokens (not <pad>, not <eos>, not the cropped first/last position) and P_c the
// softmax of channel c, that is
//     logit[i][j] = m_i m_j (A[i][j] + A[j][i]) - sum_c (w_c / t_c) r_c[i] r_c[j] + bias
//     A = sum_c w_c P_c,     r_c[i] = m_i (sum_j m_j P_c[i][j] + sum_j m_j P_c[j][i]),     t_c = sum_i r_c[i]
// so one [T,T] fp32 accumulator per sequence plus the masked row and column sums of every channel are enough.
// `model.predict_contacts(tokens)` (esm2.py:146-147) returns only the map, so the engine takes this path for
// it; `forward(return_contacts=True)` also returns "attentions" and keeps the materialised path
// (elementwise.hip: contact_sums_kernel / contact_out_kernel).
//
// Per layer, after the flash attention kernel has left q, k (scaled, rotated) and the row log-sum-exp:
//   contact_accum_kernel  re-computes S = q k^T in 32x32 MFMA blocks, P = exp(S + key_bias - lse), and adds
//                         w_c P into A, partial masked row / column sums into rowp / colp (no atomics: every
//                         output element has exactly one writer, so the result is deterministic);
//   contact_reduce_kernel sums the partials in a fixed order into rowsum[b,c,:] and colsum[b,c,:].
// After the last layer: contact_rt_kernel (r_c, w_c / t_c) and contact_final_kernel (the formula above).
#include "common.h"
#include "kernels.h"

namespace esmk {

namespace {

template <int CTRL>
ESMK_DEV float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
ESMK_DEV float row16_sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}

// v[lane] + v[lane ^ 32] in every lane: v_permlane32_swap_b32 exchanges the upper half of its first operand with
// the lower half of its second one (a VALU op; __shfl_xor goes through the LDS crossbar and an lgkmcnt wait)
ESMK_DEV float half_swap_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// residue mask of token position t of one sequence (modules.py:340-347 + the pad mask of esm2.py:135-139)
ESMK_DEV float residue_mask(const int64_t* tok, int t, int Tlen, int pad_idx, int eos_idx, int bos, int eos) {
    if (t < bos || t >= Tlen - eos) return 0.f;
    const long long v = tok[t];
    return (v == pad_idx || (eos && v == eos_idx)) ? 0.f : 1.f;
}

}  // namespace

// grid: G * B * ceil(T/128) * ceil(T/128) workgroups of 4 waves = (head group g, sequence b, 128-key chunk,
// 128-query block); wave w owns queries [q0, q0 + 32), q0 = 128 * qblock + 32 w, times the chunk's 128 keys
// (4 accumulator blocks = 64 VGPRs) with the group's heads as the loop, so the w_c-weighted sum over those heads
// is formed in registers and A[g] is read-modified-written once per layer by exactly one wave per element.
// Workgroup ids are XCD-remapped so that the query blocks of one (b, chunk) — which read the same K rows — share
// an L2.  MFMA orientation as in attn_probs_kernel: D[row = query][col = key], lane & 31 = key, register = query
// row.  Masks are folded into the exponent: a masked query row gets lse = +inf, a masked key bias = -inf, so
// their probabilities are exact zeros in A and in both sums (the final formula multiplies by m_i m_j anyway).
// K fragments are fetched one block ahead (register double buffer) to cover the L2 latency.  The partial sums
// are parked in wave-private LDS and written after the last load: on gfx9 stores share the in-order vmcnt
// counter with loads, so a store between two prefetches puts its HBM write-acknowledge latency (~2 us) on the
// critical path of every block (measured: 71 % of the wave cycles parked in s_waitcnt, profiles/r1_v19_*).
// Outputs per layer: A[g] += sum_{h in g} w P_h;  rowp[b, chunk, h, q] partial row sums;  colp[b, q0/32, h, key]
// partial column sums.  Every element has one writer.
template <typename T, int HD>
__global__ __launch_bounds__(256, HD == 128 ? 1 : 2) void contact_accum_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const float* __restrict__ lse,
    const float* __restrict__ key_bias, const int64_t* __restrict__ tokens, const float* __restrict__ wreg,
    float* __restrict__ acc_out, float* __restrict__ rowp, float* __restrict__ colp, int B, int H, int Tlen,
    int layer, int G, int pad_idx, int eos_idx, int bos, int eos) {
    // per wave and head of the group: lse (log2 domain, as q.k is: attention.hip) of the 32 queries, 128 column sums, 2 x 32 row sums
    extern __shared__ float s_dyn[];
    using V8 = typename Op<T>::v8;
    constexpr int KS = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = lane >> 5, lm = lane & 31;
    const int nQ = (Tlen + 127) >> 7;  // query blocks == key chunks
    // (g, b, chunk, qblock), qblock fastest, contiguous id ranges per XCD
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = id % nQ;
    id /= nQ;
    const int kci = id % nQ;
    id /= nQ;
    const int b = id % B, g = id / B;
    const int hg = (H + G - 1) / G;  // heads per group
    const int h0 = g * hg, h1 = min(H, h0 + hg);
    const int q0 = qb * 128 + wave * 32, kc = kci * 128;
    if (q0 >= Tlen || h0 >= h1) return;  // no barriers below: waves are independent
    const int nP = (Tlen + 31) >> 5;
    float* s_lse = s_dyn + wave * (hg * 224);
    float* s_col = s_lse + hg * 32;
    float* s_row = s_col + hg * 128;
    const int64_t* tok = tokens + (size_t)b * Tlen;

    for (int idx = lane; idx < (h1 - h0) * 32; idx += 64) {
        const int hd = h0 + (idx >> 5), qq = q0 + (idx & 31);
        const bool keep = qq < Tlen && residue_mask(tok, qq, Tlen, pad_idx, eos_idx, bos, eos) != 0.f;
        s_lse[idx] = keep ? lse[((size_t)b * H + hd) * Tlen + qq] : __builtin_inff();  // lse is log2-domain
    }
    float kb2[4];  // key bias in the exp2 domain: 0, or -inf for <pad> / masked / out-of-range keys
    int krow[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int key = kc + jj * 32 + lm;
        krow[jj] = min(key, Tlen - 1);
        const bool keep = key < Tlen && residue_mask(tok, key, Tlen, pad_idx, eos_idx, bos, eos) != 0.f;
        kb2[jj] = keep ? (key_bias != nullptr ? key_bias[(size_t)b * Tlen + key] : 0.f) : -__builtin_inff();
    }
    const int qr = min(q0 + lm, Tlen - 1);
    const int nblk = min(4, (Tlen - kc + 31) >> 5);  // key blocks of this chunk that hold a key (wave uniform)

    f32x16 acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;

    auto load_k = [&](V8 (&kf)[KS], int hd, int jj) {
        const T* kp = k + (((size_t)b * H + hd) * Tlen + krow[jj]) * HD + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const V8*>(kp + 16 * ks);
    };
    auto load_q = [&](V8 (&qf)[KS], int hd) {
        const T* qp = q + (((size_t)b * H + hd) * Tlen + qr) * HD + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + 16 * ks);
    };
    V8 qf[KS], qn[KS], kf[KS], kn[KS];
    load_q(qf, h0);
    load_k(kf, h0, 0);
    for (int hd = h0; hd < h1; ++hd) {
        const int hn = min(hd + 1, h1 - 1);
        load_q(qn, hn);  // next head's queries: a whole head of lead time
        float cr[16], rs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            cr[r] = -s_lse[(hd - h0) * 32 + mfma32_row(r, hh)];
            rs[r] = 0.f;
        }
        const float wl = wreg[layer * H + hd];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            // fetch the next block's K rows (next head's block 0 after the last block)
            if (jj + 1 < 4) load_k(kn, hd, jj + 1);
            else load_k(kn, hn, 0);
            if (jj < nblk) {
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) s = Op<T>::mma(qf[ks], kf[ks], s);
                float cs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // 2^(s + key_bias - lse) with log2-domain scores; -inf + (-(+inf)) stays -inf -> 0
                    const float p = __builtin_amdgcn_exp2f(s[r] + cr[r] + kb2[jj]);
                    acc[jj][r] = __builtin_fmaf(wl, p, acc[jj][r]);
                    rs[r] += p;
                    cs += p;
                }
                cs += __shfl_xor(cs, 32, 64);  // the two lane halves hold different query rows of one key
                if (hh == 0) s_col[(hd - h0) * 128 + jj * 32 + lm] = cs;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[ks] = kn[ks];
        }
        // row sums over this chunk's keys: reduce over the 16 lanes of each DPP row; the two rows of a lane half
        // go to two partial slots
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = row16_sum(rs[r]);
            if ((lane & 15) == 0) s_row[((hd - h0) * 2 + ((lane >> 4) & 1)) * 32 + mfma32_row(r, hh)] = v;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
    }
    // every load of the main loop has retired: now the stores
    for (int idx = lane; idx < (h1 - h0) * 128; idx += 64) {
        const int key = kc + (idx & 127);
        if (key < Tlen) colp[(((size_t)b * nP + (q0 >> 5)) * H + h0 + (idx >> 7)) * Tlen + key] = s_col[idx];
    }
    for (int idx = lane; idx < (h1 - h0) * 32; idx += 64) {  // the two 16-lane slots of a row are summed here
        const int qq = q0 + (idx & 31), hl = idx >> 5;
        if (qq < Tlen)
            rowp[(((size_t)b * nQ + kci) * H + h0 + hl) * Tlen + qq] =
                s_row[(hl * 2) * 32 + (idx & 31)] + s_row[(hl * 2 + 1) * 32 + (idx & 31)];
    }
    float* A = acc_out + ((size_t)g * B + b) * Tlen * Tlen;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int key = kc + jj * 32 + lm;
        if (key < Tlen) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = q0 + mfma32_row(r, hh);
                if (qrow < Tlen) {
                    float* a = A + (size_t)qrow * Tlen + key;
                    *a = (layer == 0 ? 0.f : *a) + acc[jj][r];
                }
            }
        }
    }
}

// head_dim 64 (every model but the 15B): the same computation with the K chunk of a head staged ONCE per
// workgroup through the LDS.  The direct fragment loads above touch 32 different 128-byte lines per instruction
// (a lane reads 16 bytes of "its" key row), which makes the texture-address path the bottleneck (measured: 71 % of
// the wave cycles parked, 1.2 ms per layer at B = 64 where the VALU work is 0.2 ms).  Here 256 threads copy the
// chunk's 128 rows x 128 B with 4 coalesced global_load_lds each (XOR-swizzled 16-byte slots, the layout of
// attn_fwd_kernel), double buffered across heads with one barrier per head; fragments come from ds_read_b128.
// The partial sums are parked per slab of PARK heads, so the LDS budget (67 KiB: two workgroups per CU) does not
// depend on the number of heads.
template <typename T>
__global__ __launch_bounds__(256, 2) void contact_accum64_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const float* __restrict__ lse,
    const float* __restrict__ key_bias, const int64_t* __restrict__ tokens, const float* __restrict__ wreg,
    float* __restrict__ acc_out, float* __restrict__ rowp, float* __restrict__ colp, int B, int H, int Tlen,
    int layer, int G, int pad_idx, int eos_idx, int bos, int eos) {
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    using V8 = typename Op<T>::v8;
    constexpr int PARK = 10, KBUF = 128 * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, lm = lane & 31;
    const int nQ = (Tlen + 127) >> 7;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = id % nQ;
    id /= nQ;
    const int kci = id % nQ;
    id /= nQ;
    const int b = id % B, g = id / B;
    const int hg = (H + G - 1) / G;
    const int h0 = g * hg, h1 = min(H, h0 + hg);
    if (h0 >= h1) return;  // workgroup uniform
    const int q0 = qb * 128 + wave * 32, kc = kci * 128;
    const bool active = q0 < Tlen;  // wave uniform; inactive waves still stage K and meet the barriers
    const int nP = (Tlen + 31) >> 5;
    char* s_k = s_raw;
    float* s_lse = reinterpret_cast<float*>(s_raw + 2 * KBUF) + wave * (PARK * 224);
    float* s_col = s_lse + PARK * 32;
    float* s_row = s_col + PARK * 128;
    const int64_t* tok = tokens + (size_t)b * Tlen;

    // K staging: position pos = 256 j + tid of the chunk image = (row pos / 8, 16-byte slot pos % 8)
    const T* kbase = k + (size_t)b * H * Tlen * 64;
    size_t ksrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pos = j * 256 + tid;
        const int r = pos >> 3, sl = pos & 7;
        ksrc[j] = (size_t)min(kc + r, Tlen - 1) * 64 + (sl ^ ((r >> 1) & 7)) * 8;
    }
    auto stage = [&](int buf, int hd) {
        const T* kh = kbase + (size_t)hd * Tlen * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(kh + ksrc[j], s_k + buf * KBUF + (j * 256 + wave * 64) * 16);
    };
    int xo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xo[ks] = lm * 128 + (((2 * ks + hh) ^ ((lane >> 1) & 7)) << 4);

    // key bias (0 / -inf for <pad>), -inf for masked and out-of-range keys: the score accumulators START from it,
    // so the MFMA chain delivers s + bias and the masking costs no VALU work
    float kbr[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int key = kc + jj * 32 + lm;
        const bool keep = key < Tlen && residue_mask(tok, key, Tlen, pad_idx, eos_idx, bos, eos) != 0.f;
        kbr[jj] = keep ? (key_bias != nullptr ? key_bias[(size_t)b * Tlen + key] : 0.f) : -__builtin_inff();
    }
    const int qr = min(q0 + lm, Tlen - 1);
    const int nblk = min(4, (Tlen - kc + 31) >> 5);
    const bool qkeep = q0 + lm < Tlen && residue_mask(tok, q0 + lm, Tlen, pad_idx, eos_idx, bos, eos) != 0.f;
    // does the wave's 128-key chunk hold a masked key (<cls>, <eos>, a pad, the sequence end)?  wave uniform, the same
    // for every head: 6 of the 8 chunks of a full-length sequence do not
    const unsigned masked_blocks = __builtin_amdgcn_readfirstlane(
        __builtin_amdgcn_ballot_w64(kbr[0] != 0.f || kbr[1] != 0.f || kbr[2] != 0.f || kbr[3] != 0.f) != 0 ? 1u : 0u);

    // fp32 VALU instructions take 4 cycles per wave on gfx950 (PMC: 4.5 cycles per VALU instruction in this kernel)
    // and the packed forms process two values in the same 4: the per-score arithmetic is written on float pairs
    f32x2 acc[4][8];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[jj][i] = f32x2{0.f, 0.f};
    auto load_q = [&](V8 (&qf)[4], int hd) {
        const T* qp = q + (((size_t)b * H + hd) * Tlen + qr) * 64 + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + 16 * ks);
    };
    V8 qf[4], qn[4];
    stage(0, h0);
    load_q(qf, h0);
    int cur = 0;
    for (int hs = h0; hs < h1; hs += PARK) {  // slab of heads whose partial sums are parked in the LDS
        const int he = min(h1, hs + PARK);
        if (active) {
            for (int idx = lane; idx < (he - hs) * 32; idx += 64) {
                // lane idx & 31 == lm for both halves: qkeep is the mask of query q0 + (idx & 31)
                const int qq = min(q0 + (idx & 31), Tlen - 1);
                const float v = lse[((size_t)b * H + hs + (idx >> 5)) * Tlen + qq];  // log2 domain
                s_lse[idx] = qkeep ? -v : -__builtin_inff();  // stored NEGATED: it is the start value of the score accumulators
            }
        }
        for (int hd = hs; hd < he; ++hd) {
            wait_vmcnt0();    // this thread's share of head hd's chunk has landed
            __syncthreads();  // ... everybody's has, and nobody reads the other buffer any more
            if (hd + 1 < h1) stage(cur ^ 1, hd + 1);
            if (active) {
                const char* sk = s_k + cur * KBUF;
                load_q(qn, min(hd + 1, h1 - 1));
                // -lse of the 16 query rows this lane holds (a masked query: -inf, its row comes out as exact zeros).  The
                // score accumulators of every block START from it — one set of 16 registers per head, kept intact by
                // the early-clobber MFMA of common.h — so a score leaves the matrix pipe as s - lse: no 16 v_mov per
                // block to seed the accumulator, no packed add per score pair.  The key bias (0 / -inf) is added
                // afterwards, and only by waves whose chunk holds a masked key.
                f32x16 nlse;
                f32x2 rs[8];
#pragma unroll
                for (int r = 0; r < 16; ++r) nlse[r] = s_lse[(hd - hs) * 32 + mfma32_row(r, hh)];
#pragma unroll
                for (int i = 0; i < 8; ++i) rs[i] = f32x2{0.f, 0.f};
                const float wl = wreg[layer * H + hd];
                const f32x2 wl2 = f32x2{wl, wl};
                // two copies of the block loop, chosen per wave: hipcc turns a per-block "add the key bias if the block
                // has a masked key" into 16 v_cndmask per block, which costs more than it saves
                auto blocks = [&](auto any_masked_c) {
                    constexpr bool ANY_MASKED = decltype(any_masked_c)::value;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (jj < nblk) {  // wave uniform
                            f32x16 s = Op<T>::mma_keep_c(qf[0], *reinterpret_cast<const V8*>(sk + jj * 4096 + xo[0]), nlse);
#pragma unroll
                            for (int ks = 1; ks < 4; ++ks) {
                                const V8 kf = *reinterpret_cast<const V8*>(sk + jj * 4096 + xo[ks]);
                                s = Op<T>::mma(qf[ks], kf, s);
                            }
                            f32x2 cs2 = f32x2{0.f, 0.f};
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                // 2^(s + key_bias - lse) with log2-domain scores; -inf stays -inf -> 0
                                f32x2 t = f32x2{s[2 * i], s[2 * i + 1]};
                                if constexpr (ANY_MASKED) t += f32x2{kbr[jj], kbr[jj]};
                                const f32x2 p = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                                acc[jj][i] = __builtin_elementwise_fma(wl2, p, acc[jj][i]);
                                rs[i] += p;
                                cs2 += p;
                            }
                            const float cs = half_swap_sum(cs2[0] + cs2[1]);  // the lane halves hold different query rows
                            if (hh == 0) s_col[(hd - hs) * 128 + jj * 32 + lm] = cs;
                        }
                    }
                };
                if (masked_blocks == 0) blocks(std::false_type{});
                else blocks(std::true_type{});
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = row16_sum(rs[r >> 1][r & 1]);
                    if ((lane & 15) == 0) s_row[((hd - hs) * 2 + ((lane >> 4) & 1)) * 32 + mfma32_row(r, hh)] = v;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
            }
            cur ^= 1;
        }
        if (active) {
            for (int idx = lane; idx < (he - hs) * 128; idx += 64) {
                const int key = kc + (idx & 127);
                if (key < Tlen) colp[(((size_t)b * nP + (q0 >> 5)) * H + hs + (idx >> 7)) * Tlen + key] = s_col[idx];
            }
            for (int idx = lane; idx < (he - hs) * 32; idx += 64) {  // the two 16-lane slots of a row are summed here
                const int qq = q0 + (idx & 31), hl = idx >> 5;
                if (qq < Tlen)
                    rowp[(((size_t)b * nQ + kci) * H + hs + hl) * Tlen + qq] =
                        s_row[(hl * 2) * 32 + (idx & 31)] + s_row[(hl * 2 + 1) * 32 + (idx & 31)];
            }
        }
    }
    if (!active) return;
    float* A = acc_out + ((size_t)g * B + b) * Tlen * Tlen;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int key = kc + jj * 32 + lm;
        if (key < Tlen) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = q0 + mfma32_row(r, hh);
                if (qrow < Tlen) {
                    float* a = A + (size_t)qrow * Tlen + key;
                    *a = (layer == 0 ? 0.f : *a) + acc[jj][r >> 1][r & 1];
                }
            }
        }
    }
}

// rowsum[b, layer*H + h, t] = sum of the ceil(T/128) row partials, colsum[...] = sum of the ceil(T/32) column
// partials, both in index order (deterministic)
__global__ __launch_bounds__(256) void contact_reduce_kernel(const float* __restrict__ rowp,
                                                              const float* __restrict__ colp,
                                                              float* __restrict__ rowsum, float* __restrict__ colsum,
                                                              int H, int Tlen, int C, int layer, int nR, int nP) {
    const int nKb = (Tlen + 255) >> 8;
    const int bh = blockIdx.x / nKb;
    const int t = (blockIdx.x - bh * nKb) * 256 + threadIdx.x;
    if (t >= Tlen) return;
    const int b = bh / H, hd = bh - b * H;
    float s = 0.f;
    for (int p = 0; p < nR; ++p) s += rowp[(((size_t)b * nR + p) * H + hd) * Tlen + t];
    rowsum[((size_t)b * C + layer * H + hd) * Tlen + t] = s;
    s = 0.f;
    for (int p = 0; p < nP; ++p) s += colp[(((size_t)b * nP + p) * H + hd) * Tlen + t];
    colsum[((size_t)b * C + layer * H + hd) * Tlen + t] = s;
}

// one workgroup per (b, c): r = m (rowsum + colsum) written over rowsum, wt[b,c] = w_c / sum_i r[i]
__global__ __launch_bounds__(256) void contact_rt_kernel(float* __restrict__ rowsum, const float* __restrict__ colsum,
                                                          const int64_t* __restrict__ tokens,
                                                          const float* __restrict__ wreg, float* __restrict__ wt,
                                                          int C, int Tlen, int pad_idx, int eos_idx, int bos, int eos) {
    __shared__ float s_w[4];
    const int bc = blockIdx.x;
    const int b = bc / C, c = bc - b * C;
    const int64_t* tok = tokens + (size_t)b * Tlen;
    float tot = 0.f;
    for (int i = threadIdx.x; i < Tlen; i += 256) {
        const size_t o = (size_t)bc * Tlen + i;
        const float r = residue_mask(tok, i, Tlen, pad_idx, eos_idx, bos, eos) * (rowsum[o] + colsum[o]);
        rowsum[o] = r;
        tot += r;
    }
    tot = wave_sum(tot);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = tot;
    __syncthreads();
    // no residue at all: 0/0 like the reference's apc (modules.py:36-38)
    if (threadIdx.x == 0) wt[bc] = wreg[c] / ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

// 32 x 32 output tile per workgroup (cropped coordinates i, j in [0,S); token position = i + bos).  A is the sum of
// the G head-group accumulators; the rank-C apc term runs over LDS-staged slabs of 32 channels.
__global__ __launch_bounds__(256) void contact_final_kernel(const float* __restrict__ acc,
                                                             const float* __restrict__ r,
                                                             const float* __restrict__ wt,
                                                             const int64_t* __restrict__ tokens,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int B, int G, int C, int Tlen, int pad_idx, int eos_idx,
                                                             int bos, int eos) {
    __shared__ float s_t[32][33];
    __shared__ float s_ri[32][33], s_rj[32][33];
    const int S = Tlen - bos - eos;
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t* tok = tokens + (size_t)b * Tlen;
    const size_t gstride = (size_t)B * Tlen * Tlen;
    const float* A = acc + (size_t)b * Tlen * Tlen;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {  // mirrored tile A[j][i], read coalesced along i
        const int j = j0 + ty + 8 * kq, i = i0 + tx;
        float v = 0.f;
        if (j < S && i < S)
            for (int g = 0; g < G; ++g) v += A[g * gstride + (size_t)(j + bos) * Tlen + i + bos];
        s_t[ty + 8 * kq][tx] = v;
    }
    __syncthreads();
    const int jc = min(j0 + tx, S - 1) + bos;
    const int it = min(i0 + tx, S - 1) + bos;
    const float mj = (j0 + tx < S) ? residue_mask(tok, jc, Tlen, pad_idx, eos_idx, bos, eos) : 0.f;
    float sym[4], apc[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int i = i0 + ty + 8 * kq;
        const int ic = min(i, S - 1) + bos;
        const float mi = (i < S) ? residue_mask(tok, ic, Tlen, pad_idx, eos_idx, bos, eos) : 0.f;
        float v = 0.f;
        if (i < S && j0 + tx < S)
            for (int g = 0; g < G; ++g) v += A[g * gstride + (size_t)ic * Tlen + jc];
        sym[kq] = (v + s_t[tx][ty + 8 * kq]) * mi * mj;
        apc[kq] = 0.f;
    }
    const float* rb = r + (size_t)b * C * Tlen;
    const float* wb = wt + (size_t)b * C;
    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {  // channel c0 + ty + 8 kq: r[i0 + tx] and (w/t) r[j0 + tx]
            const int c = c0 + ty + 8 * kq;
            const bool ok = c < C;
            const float* rc = rb + (size_t)(ok ? c : 0) * Tlen;
            s_ri[ty + 8 * kq][tx] = ok ? rc[it] : 0.f;
            s_rj[ty + 8 * kq][tx] = ok ? wb[c] * rc[jc] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < 32; ++cc) {
            const float f = s_rj[cc][tx];
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) apc[kq] += f * s_ri[cc][ty + 8 * kq];
        }
    }
    const float bb = bias ? bias[0] : 0.f;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int i = i0 + ty + 8 * kq, j = j0 + tx;
        if (i < S && j < S) {
            const float z = sym[kq] - apc[kq] + bb;
            out[((size_t)b * S + i) * S + j] = 1.0f / (1.0f + expf(-z));
        }
    }
}

// head groups: enough workgroups to fill 256 CUs a few times over even for one short sequence
int contacts_head_groups(int B, int T, int H, int head_dim) {
    const long long nQ = (T + 127) / 128;
    const long long per_group = (long long)B * nQ * nQ;
    long long G = (1024 + per_group - 1) / per_group;
    if (G > H) G = H;
    if (G < 1) G = 1;
    int hg = (int)((H + G - 1) / G);
    if (head_dim == 128 && hg > 20) hg = 20;  // direct-load kernel: 4 waves x 224 LDS floats per head of the group
    return (H + hg - 1) / hg;  // groups that actually hold a head
}

template <typename T, int HD>
static hipError_t launch_accum(const void* q, const void* k, const float* lse, const float* key_bias,
                               const int64_t* tokens, const float* wreg, float* acc, float* rowp, float* colp,
                               int B, int H, int T_, int layer, int G, int pad_idx, int eos_idx, int bos, int eos,
                               hipStream_t st) {
    const int hg = (H + G - 1) / G;
    const long long nQ = (T_ + 127) / 128;
    const long long grid = (long long)G * B * nQ * nQ;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    static bool attr_set = false;
    if constexpr (HD == 64) {
        constexpr size_t lds = 2 * 128 * 128 + 4 * 10 * 224 * sizeof(float);  // K double buffer + parked sums
        auto kern = contact_accum64_kernel<T>;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, (const T*)q, (const T*)k, lse, key_bias,
                           tokens, wreg, acc, rowp, colp, B, H, T_, layer, G, pad_idx, eos_idx, bos, eos);
    } else {
        const size_t lds = (size_t)4 * hg * 224 * sizeof(float);
        if (hg > 20) return hipErrorInvalidValue;  // contacts_head_groups() keeps groups at <= 20 heads (70 KiB)
        auto kern = contact_accum_kernel<T, HD>;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 20 * 224 * 4);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, (const T*)q, (const T*)k, lse, key_bias,
                           tokens, wreg, acc, rowp, colp, B, H, T_, layer, G, pad_idx, eos_idx, bos, eos);
    }
    return hipGetLastError();
}

hipError_t launch_contacts_fused_layer(const void* q, const void* k, const float* lse, const float* key_bias,
                                       const int64_t* tokens, const float* wreg, float* acc, float* rowsum,
                                       float* colsum, float* rowp, float* colp, int B, int H, int T, int C, int layer,
                                       int head_dim, int pad_idx, int eos_idx, int prepend_bos, int append_eos,
                                       int operand_dtype, hipStream_t st) {
    const int bos = prepend_bos ? 1 : 0, eos = append_eos ? 1 : 0;
    const int G = contacts_head_groups(B, T, H, head_dim);
    hipError_t e;
    if (operand_dtype == ESMK_DT_BF16) {
        e = head_dim == 128 ? launch_accum<__bf16, 128>(q, k, lse, key_bias, tokens, wreg, acc, rowp, colp, B, H, T, layer,
                                                        G, pad_idx, eos_idx, bos, eos, st)
                            : launch_accum<__bf16, 64>(q, k, lse, key_bias, tokens, wreg, acc, rowp, colp, B, H, T, layer,
                                                       G, pad_idx, eos_idx, bos, eos, st);
    } else {
        e = head_dim == 128 ? launch_accum<_Float16, 128>(q, k, lse, key_bias, tokens, wreg, acc, rowp, colp, B, H, T,
                                                          layer, G, pad_idx, eos_idx, bos, eos, st)
                            : launch_accum<_Float16, 64>(q, k, lse, key_bias, tokens, wreg, acc, rowp, colp, B, H, T,
                                                         layer, G, pad_idx, eos_idx, bos, eos, st);
    }
    if (e != hipSuccess) return e;
    const int nP = (T + 31) / 32, nR = (T + 127) / 128;
    const unsigned grid = (unsigned)(B * H) * (unsigned)((T + 255) / 256);
    hipLaunchKernelGGL(contact_reduce_kernel, dim3(grid), dim3(256), 0, st, rowp, colp, rowsum, colsum, H, T, C, layer,
                       nR, nP);
    return hipGetLastError();
}

hipError_t launch_contacts_fused_final(const float* acc, float* rowsum, const float* colsum, float* wt,
                                       const int64_t* tokens, const float* wreg, const float* bias, float* out,
                                       int B, int H, int C, int T, int head_dim, int pad_idx, int eos_idx,
                                       int prepend_bos, int append_eos, hipStream_t st) {
    const int bos = prepend_bos ? 1 : 0, eos = append_eos ? 1 : 0;
    const int S = T - bos - eos;
    if (S <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(contact_rt_kernel, dim3(B * C), dim3(256), 0, st, rowsum, colsum, tokens, wreg, wt, C, T,
                       pad_idx, eos_idx, bos, eos);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nt = (S + 31) / 32;
    hipLaunchKernelGGL(contact_final_kernel, dim3(nt, nt, B), dim3(256), 0, st, acc, rowsum, wt, tokens, bias, out, B,
                       contacts_head_groups(B, T, H, head_dim), C, T, pad_idx, eos_idx, bos, eos);
    return hipGetLastError();
}

}  // namespace esmk
