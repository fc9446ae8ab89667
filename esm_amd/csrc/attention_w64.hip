// attention_w64.hip — the head_dim-64 attention core with 64 query rows per wave (round 5).
//
// Same algorithm, same arithmetic and the same bits as attn_fwd_kernel (attention.hip; reference
// esm/multihead_attention.py:357-394): swapped S^T = K . Q^T on 32x32x16 MFMAs, log2-domain scores, lazy exponent
// offset riding in as the C operand of the first MFMA of a score tile, fp32 softmax, P from registers into the
// P . V MFMA, K / V^T tiles of 64 keys double-buffered through LDS-DMA.  What changes is the SHAPE of the work:
//
//   attn_fwd_kernel     workgroup = 4 waves x 32 query rows = 128 rows, 3 workgroups per CU (168 VGPRs)
//   attn_fwd_w64_kernel workgroup = 4 waves x 64 query rows = 256 rows, 2 workgroups per CU (<= 256 registers)
//
// A wave owns TWO 32-row query blocks (A: rows q0 .. q0+31, B: q0+32 .. q0+63).  Every K fragment and every V^T
// fragment it reads from LDS feeds two MFMAs instead of one, and a workgroup stages, waits for and synchronises on
// a K / V^T tile once per 256 query rows instead of once per 128: per flop the kernel issues half the ds_read_b128,
// half the LDS-DMA, half the barriers (the 5.9 ms of attn_fwd_kernel that is neither
// MFMA nor softmax arithmetic — profiles/r4_attention_decomposition.log).  The two blocks' MFMA chains are
// independent, so a dependent MFMA never waits for its predecessor's result.  At T = 1024 the grid is B.H x 4
// workgroups = exactly 10 rounds of 512 resident workgroups at the bench shape, where the 128-row kernel runs
// 13.3 rounds of 768 (a third-full last round).
//
// BITS: each 32-row block keeps its own offset, its own "every row has seen a finite score" flag and takes its own
// fast / exact decision with its own ballot — a block is exactly a wave of attn_fwd_kernel, with the same operations
// in the same order, so ctx and lse are bit-identical to attn_fwd_kernel's (tests/test_attention_w64_gpu.py) and
// padded == packed == alone keeps holding whichever kernel a launch picks.
#include "common.h"
#include "kernels.h"
#include <math.h>
#include <stdlib.h>

namespace esmk {

namespace {
constexpr float W64_LOG2E = 1.4426950408889634f;
constexpr float W64_LAZY_LIMIT = 4096.f;  // = attention.hip LAZY_LIMIT
constexpr int W_TILE = 64 * 128;          // bytes of one K (or V^T) tile: 64 rows x 128 B
constexpr int W_STAGE = 2 * W_TILE + 256 + 16;  // K + V^T + 64 fp32 key-bias values + "tile has a masked key" flag
}  // namespace

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_w64_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
    const float* __restrict__ key_bias, const int* __restrict__ seq_info, T* __restrict__ ctx,
    float* __restrict__ lse, int H, int BH, int nq, int Tlen, int Tp, int xcdmap, int fill_mode,
    const int* __restrict__ any_pad) {
    __shared__ __attribute__((aligned(16))) char smem[2 * W_STAGE];
    using V8 = typename Op<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lm = lane & 31;
    // workgroup id -> (batch*head, query block): ids with equal id % 8 (one XCD) share bh
    int bh, qblk;
    {
        const int id = blockIdx.x;
        const int bh8 = xcdmap ? (BH & ~7) : 0;
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
    const int q0 = qblk * 256 + wave * 64;
    const size_t rbase = (size_t)bh * Tlen;

    // padding information of this sequence (wave uniform) — as attn_fwd_kernel without the packed-batch cases
    int kv_end = Tlen;
    bool use_mask = (Tlen & 63) != 0;
    if (fill_mode) {
        if (key_bias != nullptr && any_pad != nullptr && any_pad[0] != 0) use_mask = true;
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
    const T* kb = k + rbase * 64;
    const T* vb = vt + (size_t)bh * 64 * Tp;

    // padded query rows take the last real row's place in their 32-row block; padding-only blocks do no work
    const int q_end = max(kv_end, 1);
    bool act[2];
    V8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        act[qb] = (q0 + 32 * qb) < q_end && kv_end > 0;  // wave uniform
        const int qr = min(q0 + 32 * qb + lm, q_end - 1);
        const T* qp = q + (rbase + qr) * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const V8*>(qp + 16 * ks);
    }

    // staging: identical to attn_fwd_kernel (fixed per-lane byte offsets, tile advance in the scalar offset)
    const int r0 = tid >> 3;
    const int kcol = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;
    const unsigned kvo0 = (unsigned)(r0 * 64 + kcol) * (unsigned)sizeof(T), kvo1 = kvo0 + 32u * 128u;
    const unsigned vvo0 = (unsigned)(r0 * Tp + kcol) * (unsigned)sizeof(T), vvo1 = vvo0 + 32u * (unsigned)Tp * (unsigned)sizeof(T);
    auto uniform_ptr = [](const void* p) {
        const unsigned long long a = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t kdesc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(kb), 0, __builtin_amdgcn_readfirstlane(Tlen * 128), 0x00020000);
    const size_t vbytes = (size_t)64 * Tp * sizeof(T);
    const __amdgpu_buffer_rsrc_t vdesc = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(vb), 0, __builtin_amdgcn_readfirstlane(vbytes > 0x7fffffffu ? 0x7fffffff : (int)vbytes), 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * W_STAGE;
        const int k0 = kt * 64;
        const unsigned sk_off = (unsigned)kt * 8192u, sv_off = (unsigned)kt * 128u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kdesc, (lds_ptr)(base + (wave * 64) * 16), 16, kvo0, sk_off, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vdesc, (lds_ptr)(base + W_TILE + (wave * 64) * 16), 16, vvo0, sv_off, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kdesc, (lds_ptr)(base + (256 + wave * 64) * 16), 16, kvo1, sk_off, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vdesc, (lds_ptr)(base + W_TILE + (256 + wave * 64) * 16), 16, vvo1, sv_off, 0, 0);
        if (use_mask && tid < 64) {
            const int key = k0 + tid;
            float bv = -INFINITY;
            if (key < Tlen) {
                bv = key_bias ? key_bias[(size_t)b * Tlen + key] : 0.f;
                if (fill_mode) bv = (bv != 0.f) ? INFINITY : 0.f;
            }
            reinterpret_cast<float*>(base + 2 * W_TILE)[tid] = bv;
            const bool any_masked = __builtin_amdgcn_ballot_w64(bv != 0.f) != 0;
            if (tid == 0) *reinterpret_cast<int*>(base + 2 * W_TILE + 256) = any_masked ? 1 : 0;
        }
    };

    const int lrow = lm * 128;
    const int swz = (lane >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = ((2 * c + h) ^ swz) << 4;

    f32x16 o[2][2];  // [block][dv half]
    f32x16 negm[2];  // -m_off of the block's row in every slot
    float m_off[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f};
    bool m_ok[2] = {false, false};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][0][r] = 0.f, o[qb][1][r] = 0.f, negm[qb][r] = 0.f;
    }

    if (ntiles > 0) stage(0, 0);
    wait_vmcnt0();
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* sk = smem + cur * W_STAGE;
        if (kt + 1 < ntiles) stage(cur ^ 1, kt + 1);
        cur ^= 1;
        const char* sv = sk + W_TILE;
        const float* sb = reinterpret_cast<const float*>(sk + 2 * W_TILE);
        if (act[0]) {
            const bool tile_masked =
                use_mask && __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(sk + 2 * W_TILE + 256)) != 0;
            // one 32-key half of the score tile of block qb: S^T = K . Q^T (- m_off), then the key padding / tail mask
            // (multihead_attention.py:368-374)
            auto mask_half = [&](f32x16& s, int qb, int t2) {
                const float fillv = -10000.f * W64_LOG2E - m_off[qb];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + t2 * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[4 * g + e] = (bv[e] == INFINITY) ? fillv : s[4 * g + e] + bv[e];
                }
            };
            V8 pf[2][4];
            bool ok[2];  // wave uniform: every row of the block has a finite offset -> the tile may go without touching it
            float ps[2] = {0.f, 0.f};
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) ok[qb] = act[qb] && __builtin_amdgcn_ballot_w64(!m_ok[qb]) == 0;
            // ---- fast attempt, 32 keys at a time, BOTH blocks off one read of each K fragment.  The score registers of a
            // half are dead once its exponentials are taken (64 instead of 128 live across the tile: what lets two blocks
            // fit 256 registers); a block that then needs the exact path recomputes its scores (rare: LAZY_LIMIT).
            if (ok[0] || ok[1]) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f32x16 s[2];
                    {
                        const V8 kf = *reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[0]);
                        s[0] = Op<T>::mma_keep_c(kf, qf[0][0], negm[0]);
                        s[1] = Op<T>::mma_keep_c(kf, qf[1][0], negm[1]);
                    }
#pragma unroll
                    for (int ks = 1; ks < 4; ++ks) {
                        const V8 kf = *reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[ks]);
                        s[0] = Op<T>::mma(kf, qf[0][ks], s[0]);
                        s[1] = Op<T>::mma(kf, qf[1][ks], s[1]);
                    }
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        if (!ok[qb]) continue;
                        if (tile_masked) mask_half(s[qb], qb, t2);
                        // ONE dependent chain of plain v_add_f32 per block, in attn_fwd_kernel's order (t2, ks, e)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float p = __builtin_amdgcn_exp2f(s[qb][8 * ks + e]);
                                ps[qb] += p;
                                pf[qb][2 * t2 + ks][e] = Op<T>::from(p);
                            }
                    }
                }
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (!act[qb]) continue;
                bool exact = true;
                if (ok[qb]) {
                    exact = __builtin_amdgcn_ballot_w64(!(ps[qb] <= W64_LAZY_LIMIT)) != 0;  // also catches inf / NaN
                    if (!exact) lsum[qb] += ps[qb];
                }
                if (exact) {
                    // ---- exact path of this block: its whole score tile again (the first tile of a row, spikes) --------
                    f32x16 st[2];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        st[t2] = Op<T>::mma_keep_c(*reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[0]), qf[qb][0], negm[qb]);
#pragma unroll
                        for (int ks = 1; ks < 4; ++ks) {
                            const V8 kf = *reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[ks]);
                            st[t2] = Op<T>::mma(kf, qf[qb][ks], st[t2]);
                        }
                        if (tile_masked) mask_half(st[t2], qb, t2);
                    }
                    float mx = st[0][0];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t2][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const float delta = m_ok[qb] ? fmaxf(mx, 0.f) : (mx == -INFINITY ? 0.f : mx);
                    const float alpha = m_ok[qb] ? __builtin_amdgcn_exp2f(-delta) : 1.f;
                    m_ok[qb] = m_ok[qb] || (mx != -INFINITY);
                    float pe = 0.f;
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float p = __builtin_amdgcn_exp2f(st[t2][8 * ks + e] - delta);
                                pe += p;
                                pf[qb][2 * t2 + ks][e] = Op<T>::from(p);
                            }
                    lsum[qb] = lsum[qb] * alpha + pe;
                    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                        for (int d = 0; d < 2; ++d)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[qb][d][r] *= alpha;
                    }
                    m_off[qb] += delta;
#pragma unroll
                    for (int r = 0; r < 16; ++r) negm[qb][r] = -m_off[qb];
                }
            }
            // ---- O^T += V^T . P^T, both blocks off one read of each V^T fragment --------------------------------
            {
                if (act[1]) {
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const V8 vf = *reinterpret_cast<const V8*>(sv + d * 4096 + lrow + xo[kk]);
                            o[0][d] = Op<T>::mma(vf, pf[0][kk], o[0][d]);
                            o[1][d] = Op<T>::mma(vf, pf[1][kk], o[1][d]);
                        }
                } else {
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const V8 vf = *reinterpret_cast<const V8*>(sv + d * 4096 + lrow + xo[kk]);
                            o[0][d] = Op<T>::mma(vf, pf[0][kk], o[0][d]);
                        }
                }
            }
        }  // act[0]
        wait_vmcnt0();
        __syncthreads();
    }

    // ---- normalise and store ctx[b*T + q][head*64 + dv]; each block through 4 KiB of the wave's 8 KiB LDS slice ----
    using V4 = typename Op<T>::v4;
    T* dst = ctx + (size_t)b * Tlen * ((size_t)H * 64) + head * 64;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float ltot = lsum[qb] + __shfl_xor(lsum[qb], 32, 64);
        const float inv = act[qb] ? 1.0f / ltot : 0.f;
        char* wl = smem + wave * 8192 + qb * 4096;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                V4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = Op<T>::from(o[qb][d][4 * g + e] * inv);
                *reinterpret_cast<V4*>(wl + lm * 128 + (((4 * d + g) ^ (lm & 7)) << 4) + 8 * h) = pk;
            }
        const int qb0 = q0 + 32 * qb;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pc = it * 64 + lane;
            const int r = pc >> 3, c = pc & 7;
            const V8 v = *reinterpret_cast<const V8*>(wl + r * 128 + ((c ^ (r & 7)) << 4));
            if (qb0 + r < Tlen) *reinterpret_cast<V8*>(dst + (size_t)(qb0 + r) * ((size_t)H * 64) + c * 8) = v;
        }
        const int qrow = qb0 + lm;
        if (lse != nullptr && h == 0 && qrow < Tlen) lse[rbase + qrow] = act[qb] ? m_off[qb] + log2f(ltot) : 0.f;
    }
}

hipError_t launch_attention_w64(const void* q, const void* k, const void* vt, const float* key_bias,
                                const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                                int operand_dtype, int fill_mode, const int* any_pad, hipStream_t st) {
    if (B <= 0 || H <= 0 || T <= 0 || Tp < T || (Tp & 63)) return hipErrorInvalidValue;
    const int nq = (T + 255) / 256;
    dim3 grid(nq * B * H);
#define ESMK_W64_LAUNCH(TT)                                                                                     \
    hipLaunchKernelGGL((attn_fwd_w64_kernel<TT>), grid, dim3(256), 0, st, (const TT*)q, (const TT*)k, (const TT*)vt, \
                       key_bias, seq_info, (TT*)ctx, lse, H, B * H, nq, T, Tp, 1, fill_mode, any_pad)
    if (operand_dtype == ESMK_DT_BF16) {
        ESMK_W64_LAUNCH(__bf16);
    } else if (operand_dtype == ESMK_DT_F16) {
        ESMK_W64_LAUNCH(_Float16);
    } else {
        return hipErrorInvalidValue;
    }
#undef ESMK_W64_LAUNCH
    return hipGetLastError();
}

}  // namespace esmk
