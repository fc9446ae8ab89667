// attention.hip — multi-head self-attention core for head_dim 64 on gfx950.
//
// Replaces, for one TransformerLayer, the reference's materialised attention
// (esm/multihead_attention.py:357 bmm(q,k^T) -> :368-374 key padding fill -> :379 fp32 softmax
// -> :387 bmm(probs,v) -> :394 merge heads) with a flash-style kernel: the [T,T] score matrix
// never leaves the CU.  q (already scaled and rotated), k (rotated) come from the fused QKV
// epilogue in gemm.hip as [B,H,T,64]; v comes TRANSPOSED as vt[B,H,64,Tp] with the key index
// permuted inside groups of 16 (4-groups 1 and 2 swapped).
//
// SCORE DOMAIN: q arrives scaled by d^-1/2 * log2(e) (the QKV epilogue folds it into the q scale), so q.k is
// the score in the log2 domain and every softmax here is exp2; the saved row log-sum-exp `lse` is log2-domain
// too (attn_probs_kernel, contacts.hip: p = exp2(q.k - lse)).  The masked_fill constant of the MSA column
// attention (-10000, axial_attention.py:211-215) is scaled accordingly.
//
// attn_fwd_kernel: one workgroup = 4 waves = 128 query rows of one (batch, head); each wave
// owns 32 query rows.  K / V^T tiles of 64 keys are staged HBM->LDS by global_load_lds_dwordx4
// (double buffered; a three-buffer variant with a counted vmcnt measured 6 % slower in round 1 and was removed
// together with the 64-rows-per-wave, 4-waves-per-SIMD and packed-fp32 variants — numbers in DESIGN.md §4.2;
// 128-byte rows, 16-byte chunks XOR-swizzled with (row>>1)&7 on the source address and on
// the read, so every ds_read_b128 is bank-conflict free).
// The 1-D grid is mapped so that all query blocks of one (batch, head) run on ONE XCD (workgroup id
// % 8): its K / V^T (2 x T x 128 B) is fetched into that XCD's L2 once instead of once per XCD
// (rocprofv3 FETCH_SIZE showed 5.5x the algorithmic bytes with the row-major grid).
//   S^T = K . Q^T  is computed "swapped" (MFMA A operand = K rows, B operand = Q rows): lane l
//   then holds, for query l&31, the scores of keys (r&3)+8(r>>2)+4(l>>5) — a full softmax row
//   lives in two lanes (l, l^32), so row max / row sum need one cross-lane exchange per tile.
//   O^T += V^T . P^T : the P values a lane already holds are exactly its B-operand slots when
//   the V^T A-operand uses the same (permuted) key order, so P goes registers -> MFMA with no
//   LDS round trip and no lane permutation; O^T keeps the query in the lane (l&31) so the
//   online-softmax rescale is a lane-local multiply.
// Softmax is fp32 (exp2; see LAZY below), P is rounded to the operand dtype for the PV MFMA,
// accumulation is fp32.
//
// attn_probs_kernel: re-computes S tile by tile from q, k and the saved log-sum-exp and writes
// normalised fp32 probabilities for need_head_weights=True / the contact head
// (multihead_attention.py:396-403, esm2.py:119-121,132-139) with padded rows/cols zeroed.
#include "common.h"
#include "kernels.h"
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <type_traits>

namespace esmk {

constexpr float LOG2E = 1.4426950408889634f;
// round 1, end to end in one run (profiles/r1_v6_attention_variants.log): row-major grid 16.39 ms/step,
// XCD-grouped grid 16.14 (kept), three LDS stages 17.45, split reductions 16.64 (both removed)
constexpr int ATTN_DEFAULT_VARIANT = 1;
constexpr int A_TILE = 64 * 128;  // bytes of one K (or V^T) tile: 64 rows x 128 B
constexpr int A_STAGE = 2 * A_TILE + 256 + 16;  // K + V^T + 64 fp32 key-bias values + "tile has a masked key" flag

ESMK_DEV void attn_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    // lgkmcnt(0): this wave's LDS writes (key-bias row, epilogue staging) are complete when others pass
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// LAZY = 0: textbook online softmax (running maximum updated and O^T rescaled on every key tile).
// LAZY = 1 (default): the exponent offset m_off is only moved when it has to be.  Scores arrive in the log2
// domain (the QKV epilogue folds log2(e) into the q scale), and -m_off rides into the accumulator as the C operand
// of the first QK^T MFMA of each 32-key tile, so a score leaves the matrix pipe as s - m_off and the probability is
// ONE v_exp_f32 away: no per-score subtract, no per-tile maximum, no rescale of O^T.  After the exponentials, a
// lane's partial row sum tells whether any of its scores ran more than ~12 octaves above the offset (sum > 4096,
// inf or NaN): only then — and on every tile until each row of the wave has seen a finite score — the wave takes
// the exact path (maximum, shift of the offset, rescale), which recomputes this tile's probabilities from the
// untouched score registers.  Mathematically the same softmax; P <= 4096 stays far inside fp16 / bf16 range and
// the row sum and O^T accumulate in fp32.  Per score the VALU work drops from max + fma + exp + add + cvt +
// rescale (~7 issue slots, the pole of this kernel at head_dim 64: profiles/r1_v20_attention_pmc.txt) to
// exp + add + cvt (~4).
constexpr float LAZY_LIMIT = 4096.f;

// Round-2 experiments on top of LAZY = 1, measured on the bench shape and removed again (profiles/r2_attention_*):
// alternating the two accumulators of a phase instead of two chains of four dependent MFMAs, s_setprio 1 around the
// MFMA clusters (all within +-1 %), and a software-pipelined single-wave schedule (QK^T of tile t+1 and P.V of tile
// t-1 issued in one basic block with the exponentials of tile t; 200+ VGPRs -> 2 waves per SIMD): 15.5 vs 13.1 ms
// per step.  The PMC pass shows why: VALU-port time (711 cycles per wave and tile) plus matrix-pipe time (512)
// add up to the kernel's duration — the limiter is how well the hardware interleaves the waves of a SIMD, which a
// third resident wave helps more than in-wave scheduling.  Also measured and dropped: the row sum as 16 v_dot2c_f32_f16
// against (1, 1) on the packed P instead of 32 v_add_f32 (481-487 vs 478.6 us on the micro-benchmark: a dot2c costs more
// than the two adds it replaces); unrolling the key-tile loop by two so that the LDS buffer is a compile-time constant
// (fragment addresses as lane base + immediate): 332 bytes of scratch at the 168-VGPR budget of 3 waves per SIMD; the
// row sum on the matrix pipe (4 MFMAs of an all-ones A operand with the P fragments, Q re-read from an LDS image to free
// the accumulator's 16 registers; exact, 168 VGPRs): 511 vs 452 us — the wave has to read the MFMA result back for the
// overflow check before P.V may start, and that dependency costs more than the 32 adds.
// Round 5: the same kernel with 64 query rows per wave (two 32-row blocks per wave sharing every K / V^T fragment read, one
// barrier and one LDS-DMA round per 256 query rows, 256 registers -> 2 waves per SIMD; bit-identical, git bb4a853
// attention_w64.hip): 13.9 vs 12.6 ms per step at B = 64, 1.48 vs 1.13 at B = 4 — the third resident wave is worth more than
// half the LDS traffic (profiles/r5_attention_w64_and_resid_atomic_ab.log).  Removed.
// HACK (timing experiments, results WRONG; ESMK_ATTN_HACK): 1 = no row-sum adds, 2 = no exponentials (p = score),
// 4 = no P.V MFMAs, 8 = no QK^T MFMAs — which of VALU issue and the matrix pipe the kernel's time follows.
// X3 (precision mode f16x3, esmk_config::weight_split 4): the context rows leave as the A operand of that mode's out-projection —
// per head (= one 64-column K tile) hi | hi | lo, lo = T(v - T(v)), row stride 3 H 64 (the layout the LayerNorm kernel writes
// with LnExtra::x3).  An instantiation of its own: the shipped kernel keeps its code.
template <typename T, int LAZY, bool BUF = true, int HACK = 0, bool X3 = false>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
    const float* __restrict__ key_bias, const int* __restrict__ seq_info, T* __restrict__ ctx,
    float* __restrict__ lse, int H, int BH, int nq, int Tlen, int Tp, int xcdmap, int fill_mode,
    const int* __restrict__ any_pad, AttnSegs segs, int stagger) {
    __shared__ __attribute__((aligned(16))) char smem[2 * A_STAGE];
    using V8 = typename Op<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Three workgroups share a CU, one wave of each per SIMD.  They are dispatched together and every key tile costs the
    // same, so the three waves of a SIMD run in lockstep: all in their QK^T / PV MFMAs, then all in their exponentials —
    // matrix pipe and VALU take turns instead of overlapping (PMC, DESIGN.md §4.2: VALU-port time + MFMA time = the
    // kernel's duration).  `stagger` > 0 delays a workgroup by (its wave slot % 3) x stagger cycles once, at the start:
    // the slot number (HW_ID.WAVE_ID) tells the co-resident workgroups apart.  Timing only — results do not change.
    if (stagger > 0) {
        __shared__ int s_slot;
        if (tid == 0) s_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, bits 3:0 = wave slot
        __syncthreads();
        const long long wait = (long long)(s_slot % 3) * stagger;
        if (wait > 0) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while ((long long)(__builtin_readcyclecounter() - t0) < wait) __builtin_amdgcn_s_sleep(8);
        }
    }
    const int h = lane >> 5, lm = lane & 31;
    // workgroup id -> (batch*head, query block): ids with equal id % 8 (one XCD) share bh
    int bh, qblk;
    {
        const int id = blockIdx.x;
        const int bh8 = xcdmap ? (BH & ~7) : 0;  // heads covered by the XCD-grouped part of the grid
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
    // Token-packed batches (segs.work != null; esmk_forward_packed): the "sequence" is one row space of Tlen
    // rows holding the segments back to back, and query block qblk is work item (first row of its segment,
    // segment length, first query of the block relative to the segment, segment index).  Keys, values and
    // the softmax stay inside the segment: same tiles, same order as the segment alone => same bits.
    int row0 = 0, Tseg = Tlen, qrel = qblk * 128;
    bool seg_pad = false;
    if (segs.work != nullptr) {
        row0 = __builtin_amdgcn_readfirstlane(segs.work[4 * qblk]);
        Tseg = __builtin_amdgcn_readfirstlane(segs.work[4 * qblk + 1]);
        qrel = __builtin_amdgcn_readfirstlane(segs.work[4 * qblk + 2]);
        seg_pad = segs.npad[__builtin_amdgcn_readfirstlane(segs.work[4 * qblk + 3])] > 0;
    }
    const int q0 = qrel + wave * 32;
    const size_t rbase = (size_t)bh * Tlen + row0;  // first row of the segment in the [BH, Tlen] row spaces

    // padding information of this sequence (wave uniform)
    int kv_end = Tseg;
    bool use_mask = (Tseg & 63) != 0;
    if (segs.work != nullptr) {
        if (seg_pad) use_mask = true;  // <pad> tokens inside the segment
        else key_bias = nullptr;
    } else if (fill_mode) {
        // MSA column attention: key_bias holds 0/1 fill flags, used only when the batch has a pad
        if (key_bias != nullptr && any_pad != nullptr && any_pad[0] != 0) use_mask = true;
        else key_bias = nullptr;
    } else if (key_bias != nullptr) {
        if (seq_info != nullptr) {
            if (seq_info[2 * b] > 0) {  // sequence has pads: mask them, skip all-pad tail tiles
                use_mask = true;
                kv_end = seq_info[2 * b + 1];
            }
        } else {
            use_mask = true;
        }
    }
    const int ntiles = (kv_end + 63) >> 6;

    const T* kb = k + rbase * 64;
    const T* vb = vt + (size_t)bh * 64 * Tp + row0;

    // Query rows at or past kv_end are padding (right-padded batch; kv_end = Tseg otherwise).  They take the last
    // real row's place in their wave — exactly what a wave's surplus lanes do when the sequence runs alone or
    // token-packed — and waves made of padding only do no work and write zeros.  The lazy softmax takes its
    // exact / fast decision per WAVE, so a row's bits depend on its wave mates: with this rule the mates are the
    // same in a padded batch, alone and packed, and the three layouts stay bit-identical on every real row
    // (tools/fuzz_attention_lengths.py).  The reference computes garbage on padded query rows; every caller drops it.
    const int q_end = max(kv_end, 1);
    const bool wave_active = q0 < q_end && kv_end > 0;  // wave uniform
    // Q fragments: Q[min(q0 + lm, q_end - 1)][16 ks + 8 h .. +7]
    V8 qf[4];
    {
        const int qr = min(q0 + lm, q_end - 1);
        const T* qp = q + (rbase + qr) * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + 16 * ks);
    }

    // Staging: 2 rounds of 256 lanes each for K and for V^T; lane tid copies 16-byte chunk c of row r0 (round 0)
    // and of row r0 + 32 (round 1) — the swizzle (r >> 1) & 7 is the same for both rows.  The copies are
    // buffer_load ... lds with the per-lane byte offset FIXED and the tile advance in the scalar offset: the whole
    // per-tile address arithmetic is one s_mul on the scalar unit (instruction issue, not the matrix pipe, bounds this
    // kernel — DESIGN.md §4.2 — and the global_load_lds form cost ~14 VALU instructions per tile for 64-bit
    // addresses and row clamps).  K rows past the end of the sequence / segment are out of range of the K
    // descriptor and arrive as zeros (they are masked: use_mask is set whenever the length is not a multiple of 64).
    const int r0 = tid >> 3;
    const int kcol = ((tid & 7) ^ ((r0 >> 1) & 7)) * 8;  // element offset of the chunk inside a 64-element row
    const unsigned kvo0 = (unsigned)(r0 * 64 + kcol) * (unsigned)sizeof(T), kvo1 = kvo0 + 32u * 128u;
    const unsigned vvo0 = (unsigned)(r0 * Tp + kcol) * (unsigned)sizeof(T), vvo1 = vvo0 + 32u * (unsigned)Tp * (unsigned)sizeof(T);
    auto uniform_ptr = [](const void* p) {  // provably wave-uniform for the compiler: no waterfall loop around the loads
        const unsigned long long a = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t kdesc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(kb), 0, __builtin_amdgcn_readfirstlane(Tseg * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t vdesc = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(vb), 0, __builtin_amdgcn_readfirstlane((int)(((size_t)64 * Tp - (size_t)row0) * sizeof(T) > 0x7fffffffu
                                                                     ? 0x7fffffff : (int)(((size_t)64 * Tp - (size_t)row0) * sizeof(T)))), 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * A_STAGE;
        const int k0 = kt * 64;
        const unsigned sk_off = (unsigned)kt * 8192u, sv_off = (unsigned)kt * 128u;
        if constexpr (BUF) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kdesc, (lds_ptr)(base + (wave * 64) * 16), 16, kvo0, sk_off, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vdesc, (lds_ptr)(base + A_TILE + (wave * 64) * 16), 16, vvo0, sv_off, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kdesc, (lds_ptr)(base + (256 + wave * 64) * 16), 16, kvo1, sk_off, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vdesc, (lds_ptr)(base + A_TILE + (256 + wave * 64) * 16), 16, vvo1, sv_off, 0, 0);
        } else {  // round-1 form (A/B: ESMK_ATTN bit 3): global_load_lds with per-lane 64-bit addresses, rows clamped
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kr = min(k0 + r0 + 32 * j, Tseg - 1);
                glds16(kb + (size_t)kr * 64 + kcol, base + (j * 256 + wave * 64) * 16);
                glds16(vb + (size_t)(r0 * Tp + kcol + 32 * j * Tp) + k0, base + A_TILE + (j * 256 + wave * 64) * 16);
            }
        }
        if (use_mask && tid < 64) {
            const int key = k0 + tid;
            float bv = -INFINITY;  // keys past the end of the row: excluded
            if (key < Tseg) {
                bv = key_bias ? key_bias[(size_t)b * Tlen + row0 + key] : 0.f;
                // fill flag -> +inf marker: the score is REPLACED by -10000 (masked_fill, axial_attention.py:211-215)
                if (fill_mode) bv = (bv != 0.f) ? INFINITY : 0.f;
            }
            reinterpret_cast<float*>(base + 2 * A_TILE)[tid] = bv;
            // Only tiles that contain a masked key (the tail of the row, <pad> tokens, fill flags) pay for the mask:
            // adding the 0.0 bias of an unmasked key leaves every score bit for bit unchanged, so skipping the whole
            // 80-instruction mask block on clean tiles changes nothing but the issue time.  tid < 64 is all of wave 0.
            const bool any_masked = __builtin_amdgcn_ballot_w64(bv != 0.f) != 0;
            if (tid == 0) *reinterpret_cast<int*>(base + 2 * A_TILE + 256) = any_masked ? 1 : 0;
        }
    };

    const int lrow = lm * 128;
    const int swz = (lane >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = ((2 * c + h) ^ swz) << 4;

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    // Scores are in the log2 domain.  m_off: the offset the probabilities of this row are currently expressed
    // against (finite; LAZY = 0 keeps it at the running maximum); m_ok: the row has seen a finite score.
    float m_off = 0.f;
    bool m_ok = false;
    float lsum = 0.f;  // this lane's share of the running denominator
    f32x16 negm;       // -m_off in every slot: C operand of the first MFMA of a score tile (LAZY)
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    if (ntiles > 0) stage(0, 0);
    wait_vmcnt0();
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* sk = smem + cur * A_STAGE;
        if (kt + 1 < ntiles) stage(cur ^ 1, kt + 1);
        cur ^= 1;
        const char* sv = sk + A_TILE;
        const float* sb = reinterpret_cast<const float*>(sk + 2 * A_TILE);
        if (wave_active) {

        // ---- S^T = K . Q^T (- m_off) for 64 keys (two 32-key tiles) ---------------------------
        f32x16 st[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            if constexpr (HACK & 8) {
                st[t2] = negm;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const V8 kf = *reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[ks]);
                    st[t2][ks] += (float)kf[0] * (float)qf[ks][0];
                }
            } else if constexpr (LAZY) {
                st[t2] = Op<T>::mma_keep_c(*reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[0]), qf[0], negm);
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) {
                    const V8 kf = *reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[ks]);
                    st[t2] = Op<T>::mma(kf, qf[ks], st[t2]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[t2][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const V8 kf = *reinterpret_cast<const V8*>(sk + t2 * 4096 + lrow + xo[ks]);
                    st[t2] = Op<T>::mma(kf, qf[ks], st[t2]);
                }
            }
        }
        // ---- key padding / tail mask (multihead_attention.py:368-374) -----------------------
        if (use_mask && __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(sk + 2 * A_TILE + 256)) != 0) {
            const float fillv = -10000.f * LOG2E - (LAZY ? m_off : 0.f);  // masked_fill(-10000), same domain as st
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + t2 * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        st[t2][4 * g + e] = (bv[e] == INFINITY) ? fillv : st[t2][4 * g + e] + bv[e];
                }
        }
        // ---- softmax numerators (fp32) -----------------------------------------------------
        V8 pf[4];
        bool exact = true;
        if constexpr (LAZY) {
            // wave uniform: every row has a finite offset -> try the tile without touching the offset
            if (__builtin_amdgcn_ballot_w64(!m_ok) == 0) {
                // one dependent chain of plain v_add_f32: two chains get SLP-packed into v_pk_add_f32, which costs
                // more issue time next to MFMAs than the two adds it replaces (MI355X_MICROARCH.md, price list)
                float ps = 0.f;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float p = (HACK & 2) ? st[t2][8 * ks + e] : __builtin_amdgcn_exp2f(st[t2][8 * ks + e]);
                            if constexpr (HACK & 1) {
                                if (e == 0 && ks == 0) ps += p;
                            } else {
                                ps += p;
                            }
                            pf[2 * t2 + ks][e] = Op<T>::from(p);
                        }
                exact = __builtin_amdgcn_ballot_w64(!(ps <= LAZY_LIMIT)) != 0;  // also catches inf / NaN
                if constexpr (HACK != 0) exact = false;
                if (!exact) lsum += ps;
            }
        }
        if (exact) {
            // maximum of this tile's (offset) scores over the row: 32 in this lane, 32 in lane ^ 32
            float mx = st[0][0];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t2][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // shift of the offset: up to the new maximum; the first finite tile of a row sets it outright
            float delta, alpha;
            if constexpr (LAZY) {
                delta = m_ok ? fmaxf(mx, 0.f) : (mx == -INFINITY ? 0.f : mx);
                alpha = m_ok ? __builtin_amdgcn_exp2f(-delta) : 1.f;  // nothing accumulated yet: O^T = 0, lsum = 0
            } else {
                // st holds raw scores: the offset becomes the running maximum itself
                const float m_new = m_ok ? fmaxf(m_off, mx) : (mx == -INFINITY ? 0.f : mx);
                delta = m_new;  // subtracted from the raw scores below
                alpha = m_ok ? __builtin_amdgcn_exp2f(m_off - m_new) : 1.f;
                m_off = m_new;
            }
            m_ok = m_ok || (mx != -INFINITY);
            float ps = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float p = __builtin_amdgcn_exp2f(st[t2][8 * ks + e] - delta);
                        ps += p;
                        pf[2 * t2 + ks][e] = Op<T>::from(p);
                    }
            lsum = lsum * alpha + ps;
            // the first tile of every row (alpha = 1 by definition) and exact tiles in which no row's offset moved
            // skip the 32 multiplies: multiplying by 1.0 changes no bit
            if (!LAZY || __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            if constexpr (LAZY) {
                m_off += delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m_off;
            }
        }
        // ---- O^T += V^T . P^T ----------------------------------------------------------------
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const V8 vf = *reinterpret_cast<const V8*>(sv + d * 4096 + lrow + xo[kk]);
                if constexpr (HACK & 4) o[d][kk] += (float)vf[0] * (float)pf[kk][0];
                else o[d] = Op<T>::mma(vf, pf[kk], o[d]);
            }
        }  // wave_active
        wait_vmcnt0();
        __syncthreads();
    }

    // ---- normalise and store ctx[b*T + q][head*64 + dv] ---------------------------------------
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = wave_active ? 1.0f / ltot : 0.f;  // padding-only waves: O^T is still zero
    // Stage the wave's [32 queries][64 dv] block through its private 4 KiB LDS slice (all waves
    // have passed the last barrier, the K/V buffers are dead) so that the global stores are 16 B
    // per lane and cover whole 128-byte rows instead of 8-byte pieces of 32 different rows.
    using V4 = typename Op<T>::v4;
    char* wl = smem + wave * 4096;
    constexpr int NPASS = X3 ? 2 : 1;                 // X3: the hi values, then their remainders, through the same slice
    const size_t ldc = (size_t)H * (X3 ? 192 : 64);   // context row stride in elements
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                V4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = o[d][4 * g + e] * inv;
                    const T hi = Op<T>::from(v);
                    pk[e] = pass == 0 ? hi : Op<T>::from(v - Op<T>::to(hi));
                }
                *reinterpret_cast<V4*>(wl + lm * 128 + (((4 * d + g) ^ (lm & 7)) << 4) + 8 * h) = pk;
            }
        T* dst = ctx + ((size_t)b * Tlen + row0) * ldc + head * (X3 ? 192 : 64) + (pass == 1 ? 128 : 0);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pc = it * 64 + lane;
            const int r = pc >> 3, c = pc & 7;
            const V8 v = *reinterpret_cast<const V8*>(wl + r * 128 + ((c ^ (r & 7)) << 4));
            if (q0 + r < Tseg) {
                *reinterpret_cast<V8*>(dst + (size_t)(q0 + r) * ldc + c * 8) = v;
                if constexpr (X3)
                    if (pass == 0) *reinterpret_cast<V8*>(dst + (size_t)(q0 + r) * ldc + 64 + c * 8) = v;  // hi twice
            }
        }
    }
    // log-sum-exp of the row in the log2 domain (the map kernels compute exp2(s - lse2))
    const int qrow = q0 + lm;
    if (lse != nullptr && h == 0 && qrow < Tseg) lse[rbase + qrow] = wave_active ? m_off + log2f(ltot) : 0.f;
}

// Self-test of Op<T>::mma_keep_c (common.h): the inline-asm MFMA with an early-clobber destination that keeps its C
// operand intact.  hipcc's hazard recognizer does not look inside an asm statement, so the wait states around it are
// hand counted; this kernel runs it next to the builtin on the same operands, with VALU writes of C right before it
// and a dependent MFMA right after it (the attention kernel's pattern), and stores both results and C.
template <typename T>
__global__ void mma_keep_c_selftest_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ c0,
                                           float* __restrict__ out) {
    using V8 = typename Op<T>::v8;
    const int lane = threadIdx.x;
    const V8 af = *reinterpret_cast<const V8*>(a + lane * 8), bf = *reinterpret_cast<const V8*>(b + lane * 8);
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = c0[lane * 16 + r] * 2.0f;  // VALU writes of C just before the asm MFMA
    f32x16 d_asm = Op<T>::mma_keep_c(af, bf, c);
    d_asm = Op<T>::mma(af, bf, d_asm);                              // dependent builtin MFMA (SrcC = D of the asm one)
    f32x16 d_ref = Op<T>::mma(af, bf, c);
    d_ref = Op<T>::mma(af, bf, d_ref);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        out[(0 * 64 + lane) * 16 + r] = d_asm[r];
        out[(1 * 64 + lane) * 16 + r] = d_ref[r];
        out[(2 * 64 + lane) * 16 + r] = c[r];
    }
}

hipError_t launch_mma_keep_c_selftest(const void* a, const void* b, const float* c, float* out, int operand_dtype,
                                      hipStream_t st) {
    if (operand_dtype == ESMK_DT_F16)
        hipLaunchKernelGGL(mma_keep_c_selftest_kernel<_Float16>, dim3(1), dim3(64), 0, st, (const _Float16*)a, (const _Float16*)b, c, out);
    else if (operand_dtype == ESMK_DT_BF16)
        hipLaunchKernelGGL(mma_keep_c_selftest_kernel<__bf16>, dim3(1), dim3(64), 0, st, (const __bf16*)a, (const __bf16*)b, c, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

static hipError_t launch_attention_impl(const void* q, const void* k, const void* vt, const float* key_bias,
                                        const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                                        int operand_dtype, int fill_mode, const int* any_pad, hipStream_t st,
                                        AttnSegs segs = AttnSegs(), int n_items = 0, bool x3 = false);

// start-up stagger of the co-resident workgroups in shader cycles per wave slot (attn_fwd_kernel); < 0 = read
// ESMK_ATTN_STAGGER on the first launch; esmk_debug_set("attn_stagger", cycles)
static std::atomic<int> g_attn_stagger{-1};
constexpr int kAttnStaggerDefault = 0;
void attention_set_stagger(int cycles) { g_attn_stagger = cycles < 0 ? 0 : cycles; }

hipError_t launch_attention(const void* q, const void* k, const void* vt, const float* key_bias,
                            const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                            int operand_dtype, hipStream_t st) {
    return launch_attention_impl(q, k, vt, key_bias, seq_info, ctx, lse, B, H, T, Tp, operand_dtype, 0, nullptr, st);
}

// precision mode f16x3: ctx [B*T, 3 H 64] in the hi | hi | lo layout of that mode's out-projection operand (attn_fwd_kernel X3)
hipError_t launch_attention_x3(const void* q, const void* k, const void* vt, const float* key_bias, const int* seq_info, void* ctx3,
                               float* lse, int B, int H, int T, int Tp, int operand_dtype, hipStream_t st) {
    return launch_attention_impl(q, k, vt, key_bias, seq_info, ctx3, lse, B, H, T, Tp, operand_dtype, 0, nullptr, st, AttnSegs(), 0, true);
}

hipError_t launch_attention_fill(const void* q, const void* k, const void* vt, const float* key_fill,
                                 const int* any_pad, void* ctx, float* lse, int B, int H, int T, int Tp,
                                 int operand_dtype, hipStream_t st) {
    return launch_attention_impl(q, k, vt, key_fill, nullptr, ctx, lse, B, H, T, Tp, operand_dtype, 1, any_pad, st);
}

// token-packed batch: one row space of `rows` rows, `n_items` query blocks described by segs.work
hipError_t launch_attention_packed(const void* q, const void* k, const void* vt, const float* key_bias, void* ctx,
                                   int H, int rows, int Tp, AttnSegs segs, int n_items, int operand_dtype,
                                   hipStream_t st) {
    if (segs.work == nullptr || segs.npad == nullptr || n_items <= 0) return hipErrorInvalidValue;
    return launch_attention_impl(q, k, vt, key_bias, nullptr, ctx, nullptr, 1, H, rows, Tp, operand_dtype, 0, nullptr,
                                 st, segs, n_items);
}

static hipError_t launch_attention_impl(const void* q, const void* k, const void* vt, const float* key_bias,
                                        const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                                        int operand_dtype, int fill_mode, const int* any_pad, hipStream_t st,
                                        AttnSegs segs, int n_items, bool x3) {
    if (B <= 0 || H <= 0 || T <= 0 || Tp < T || (Tp & 63)) return hipErrorInvalidValue;
    if (x3) {  // f16x3 precision mode: fp16 operands, padded batches (engine.hip)
        if (operand_dtype != ESMK_DT_F16 || segs.work != nullptr || fill_mode != 0) return hipErrorInvalidValue;
        const int nq3 = (T + 127) / 128;
        hipLaunchKernelGGL((attn_fwd_kernel<_Float16, 1, true, 0, true>), dim3(nq3 * B * H), dim3(256), 0, st, (const _Float16*)q,
                           (const _Float16*)k, (const _Float16*)vt, key_bias, seq_info, (_Float16*)ctx, lse, H, B * H, nq3, T, Tp, 1, 0,
                           any_pad, segs, 0);
        return hipGetLastError();
    }
    const int nq = segs.work != nullptr ? n_items : (T + 127) / 128;
    dim3 grid(nq * B * H);
    // ESMK_ATTN (read once): bit 0 XCD-grouped grid (default on), bit 1 = textbook online softmax instead of the
    // lazy-offset one, bit 3 = round-1 staging by global_load_lds (A/B measurements; all are exact softmax)
    static const int var = [] {
        const char* e = getenv("ESMK_ATTN");
        return e ? atoi(e) : ATTN_DEFAULT_VARIANT;
    }();
    static std::once_flag stagger_once;  // launches may come from several host threads
    std::call_once(stagger_once, [] {
        if (g_attn_stagger.load() < 0) {
            const char* e = getenv("ESMK_ATTN_STAGGER");
            g_attn_stagger = e ? atoi(e) : kAttnStaggerDefault;
        }
    });
    const int stagger = g_attn_stagger.load();
#define ESMK_ATTN_LAUNCH(TT, LZ, BF, ...)                                                                      \
    hipLaunchKernelGGL((attn_fwd_kernel<TT, LZ, BF, ##__VA_ARGS__>), grid, dim3(256), 0, st, (const TT*)q, (const TT*)k,   \
                       (const TT*)vt, key_bias, seq_info, (TT*)ctx, lse, H, B * H, nq, T, Tp, var & 1, fill_mode, any_pad, segs, stagger)
    if (operand_dtype == ESMK_DT_BF16) {
        if (var & 2) ESMK_ATTN_LAUNCH(__bf16, 0, true);
        else ESMK_ATTN_LAUNCH(__bf16, 1, true);
    } else {
#ifdef ESMK_EXPERIMENTS  // parts of the kernel removed (results wrong): experiment builds only (common.h)
        static const int hack = [] { const char* e = getenv("ESMK_ATTN_HACK"); return e ? atoi(e) : 0; }();
#else
        constexpr int hack = 0;
#endif
        if (hack != 0) {
#ifdef ESMK_EXPERIMENTS
            if (hack == 1) ESMK_ATTN_LAUNCH(_Float16, 1, true, 1);
            else if (hack == 2) ESMK_ATTN_LAUNCH(_Float16, 1, true, 2);
            else if (hack == 3) ESMK_ATTN_LAUNCH(_Float16, 1, true, 3);
            else if (hack == 4) ESMK_ATTN_LAUNCH(_Float16, 1, true, 4);
            else if (hack == 8) ESMK_ATTN_LAUNCH(_Float16, 1, true, 8);
            else if (hack == 12) ESMK_ATTN_LAUNCH(_Float16, 1, true, 12);
            else return hipErrorInvalidValue;
#endif
        } else if (var & 2) ESMK_ATTN_LAUNCH(_Float16, 0, true);
        else if (var & 8) ESMK_ATTN_LAUNCH(_Float16, 1, false);
        else ESMK_ATTN_LAUNCH(_Float16, 1, true);
    }
#undef ESMK_ATTN_LAUNCH
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// attention probabilities for need_head_weights / contacts
// ---------------------------------------------------------------------------------------------
// O = float (the reference's fp32 maps) or T (ESMK_OUT_ATTN_LOWP: `.half()` models, esmfold.py:61-67,131-135)
template <typename T, typename O = float>
__global__ __launch_bounds__(256) void attn_probs_kernel(const T* __restrict__ q,
                                                          const T* __restrict__ k,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ key_bias,
                                                          O* __restrict__ probs, int H, int Tlen,
                                                          int layer, int Ltot, int msa_C,
                                                          const int* __restrict__ any_pad) {
    // msa_C > 0: MSA column attention (axial_attention.py:207-218).  The "sequence" b is (batch, column),
    // key_bias holds 0/1 masked_fill flags used when any_pad[0] != 0, query rows are NOT zeroed and the
    // map goes to col_attentions[b_msa, layer, head, c, i, j] (msa_transformer.py:193-194).
    using V8 = typename Op<T>::v8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, lm = lane & 31;
    const int nqb = (Tlen + 127) >> 7;
    const int bh = blockIdx.x / nqb;  // 1-D grid: B*H can exceed the 65535 limit of grid.y (MSA columns)
    const int b = bh / H, head = bh - b * H;
    const int q0 = (blockIdx.x - bh * nqb) * 128 + wave * 32;
    if (q0 >= Tlen) return;
    const bool fill = msa_C > 0;
    if (fill && (any_pad == nullptr || any_pad[0] == 0)) key_bias = nullptr;

    V8 qf[4];
    {
        const int qr = min(q0 + lm, Tlen - 1);
        const T* qp = q + ((size_t)bh * Tlen + qr) * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + 16 * ks);
    }
    // per accumulator row: log-sum-exp and query-pad flag
    float row_lse[16];
    float row_keep[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = q0 + mfma32_row(r, h);
        const int qc = min(qr, Tlen - 1);
        row_lse[r] = lse[(size_t)bh * Tlen + qc];
        row_keep[r] = (!fill && key_bias != nullptr && key_bias[(size_t)b * Tlen + qc] != 0.f) ? 0.f : 1.f;
    }
    O* out = probs + (((size_t)b * Ltot + layer) * H + head) * (size_t)Tlen * Tlen;
    if (fill) {
        const int bm = b / msa_C, c = b - bm * msa_C;
        out = probs + ((((size_t)bm * Ltot + layer) * H + head) * msa_C + c) * (size_t)Tlen * Tlen;
    }

    for (int k0 = 0; k0 < Tlen; k0 += 32) {
        const int key = k0 + lm;
        const int kc = min(key, Tlen - 1);
        const T* kp = k + ((size_t)bh * Tlen + kc) * 64 + 8 * h;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const V8 kf = *reinterpret_cast<const V8*>(kp + 16 * ks);
            s = Op<T>::mma(qf[ks], kf, s);  // D[row = query][col = key]
        }
        const float kb = (key_bias != nullptr) ? key_bias[(size_t)b * Tlen + kc] : 0.f;
        if (key < Tlen) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = q0 + mfma32_row(r, h);
                if (qr < Tlen) {
                    const float sc = (fill && kb != 0.f) ? -10000.f * LOG2E : s[r] + kb;  // masked_fill vs additive -inf
                    const float p = __builtin_amdgcn_exp2f(sc - row_lse[r]) * row_keep[r];  // log2 domain
                    out[(size_t)qr * Tlen + key] = (O)p;
                }
            }
        }
    }
}

static hipError_t launch_probs_impl(const void* q, const void* k, const float* lse, const float* key_bias,
                                    float* probs, int B, int H, int T, int layer, int num_layers_total,
                                    int operand_dtype, int msa_C, const int* any_pad, hipStream_t st,
                                    bool lowp = false) {
    dim3 grid((unsigned)(((T + 127) / 128) * B * H));
    if (lowp) {  // maps in the operand dtype
        if (operand_dtype == ESMK_DT_BF16)
            hipLaunchKernelGGL((attn_probs_kernel<__bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)q,
                               (const __bf16*)k, lse, key_bias, (__bf16*)probs, H, T, layer, num_layers_total, msa_C,
                               any_pad);
        else
            hipLaunchKernelGGL((attn_probs_kernel<_Float16, _Float16>), grid, dim3(256), 0, st, (const _Float16*)q,
                               (const _Float16*)k, lse, key_bias, (_Float16*)probs, H, T, layer, num_layers_total,
                               msa_C, any_pad);
        return hipGetLastError();
    }
    if (operand_dtype == ESMK_DT_BF16)
        hipLaunchKernelGGL((attn_probs_kernel<__bf16>), grid, dim3(256), 0, st, (const __bf16*)q,
                           (const __bf16*)k, lse, key_bias, probs, H, T, layer, num_layers_total, msa_C, any_pad);
    else
        hipLaunchKernelGGL((attn_probs_kernel<_Float16>), grid, dim3(256), 0, st,
                           (const _Float16*)q, (const _Float16*)k, lse, key_bias, probs, H, T, layer,
                           num_layers_total, msa_C, any_pad);
    return hipGetLastError();
}

hipError_t launch_attention_probs(const void* q, const void* k, const float* lse,
                                  const float* key_bias, float* probs, int B, int H, int T,
                                  int layer, int num_layers_total, int operand_dtype,
                                  hipStream_t st, bool lowp) {
    return launch_probs_impl(q, k, lse, key_bias, probs, B, H, T, layer, num_layers_total, operand_dtype, 0,
                             nullptr, st, lowp);
}

hipError_t launch_attention_probs_msa(const void* q, const void* k, const float* lse, const float* key_fill,
                                      const int* any_pad, float* probs, int Bmsa, int C, int H, int R,
                                      int layer, int num_layers_total, int operand_dtype, hipStream_t st) {
    return launch_probs_impl(q, k, lse, key_fill, probs, Bmsa * C, H, R, layer, num_layers_total, operand_dtype,
                             C, any_pad, st);
}

}  // namespace esmk
