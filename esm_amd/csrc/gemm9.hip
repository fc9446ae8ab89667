// gemm9.hip — persistent nn.Linear for gfx950 with ONE wave per SIMD:
//     C[M,N] = A[M,K] . W[N,K]^T (+ bias, + fused epilogue)          K % 64 == 0, N % 8 == 0, dense operands
//
// Same contract, tile order and epilogues as gemm8.hip (reference esm/multihead_attention.py:256-261,395;
// esm/modules.py:138-139), different main loop.  gemm8 runs 8 waves (two per SIMD) on 128 x 64 wave blocks and
// hides LDS / DMA latency by ping-pong between the two waves of a SIMD: 8 barriers per K tile, 24 KiB of
// fragment reads per 32 MFMAs.  gemm9 runs 4 waves (one per SIMD, up to 512 registers each) on 128 x 128 wave
// blocks of the same 256 x 256 x 64 tile:
//   * fragment reads per MFMA drop by a third (32 KiB per 64 MFMAs per wave: 128 KiB instead of 192 KiB of
//     LDS reads per K tile and CU);
//   * ONE barrier per K tile: the wave's own instruction stream interleaves 64 MFMAs with 32 ds_read_b128
//     (fragments of the next 16-wide K sub-step, double buffered in registers) and its 16 LDS-DMA pieces;
//   * the 256 spare registers make the fp32 residual epilogue a deep software pipeline (8 pieces of the
//     residual tile in flight instead of one: the epilogue was latency bound on those loads).
// Every output element sees the same MFMA sequence over K as in gemm8 (bias enters as the C operand of the
// first MFMA, then K ascending), so results are bit-identical to gemm8's.
//
// LDS (160 KiB): two K-tile buffers of 64 KiB (A rows 0..255, then W rows 0..255; 128-byte rows, 16-byte
// chunk index XOR-swizzled with (row >> 1) & 7 on the DMA source address and on the ds_read_b128), then
// 4 x 8 KiB wave-private epilogue slices.  Wave w stages rows [128 w, 128 w + 128) of the buffer (waves 0,1:
// activations, waves 2,3: weights) as 16 pieces of 8 rows; wave (wr, wc) = (w >> 1, w & 1) computes rows
// [128 wr, +128) x columns [128 wc, +128) of the tile.
//
// One K tile (stream position s, LDS buffer cur = s & 1), ks = 16-wide K sub-step, fragment sets alternate:
//     ks 0   reads ks 1 -> set 1   DMA: second part of position s+1 -> cur^1      16 MFMA (set 0)
//     ks 1   reads ks 2 -> set 0   [DMA: third part, schedule B]                  16 MFMA (set 1)
//     ks 2   reads ks 3 -> set 1                                                  16 MFMA (set 0)
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier       (a) every wave's pieces of position s+1 have landed,
//                                                    (b) every wave has finished reading buffer cur
//     advance the DMA stream to position s+2
//     ks 3   reads ks 0 of position s+1 (buffer cur^1) -> set 0
//            DMA: first part of position s+2 -> cur                               16 MFMA (set 1)
// The vmcnt(0) never waits for anything younger than 32 MFMA slots (schedule A: 8 + 8 pieces; schedule B
// spreads them 6 + 5 + 5 and the youngest piece is 16 slots old).  The K tiles of all tiles of a workgroup
// form one stream, as in gemm8: the first operands of the next tile land during the epilogue.
#include "gemm_epi.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace esmk {

constexpr int Q_BUF = 65536;                 // one K tile: 512 rows x 128 B
constexpr int Q_WOFF = 32768;                // W rows start here inside a buffer
constexpr int Q_EPI = 2 * Q_BUF;             // wave-private epilogue slices
constexpr int Q_SLICE = 8192;
constexpr int Q_LDS = Q_EPI + 4 * Q_SLICE;   // 160 KiB

// --------------------------------------------------------------------------------------------
// fp32 epilogue of the wave's 128 x 128 block: 16 pieces of 32 rows x 32 columns through the wave's LDS
// slice (every global access covers whole 128-byte row segments).  EPI_RESID_F32 keeps D pieces of the
// residual tile in flight.  Same arithmetic as epilogue8 (old + value), so bit-identical results.
// --------------------------------------------------------------------------------------------
template <typename T, int EPI, bool FULL, int D = 8>
ESMK_DEV void epilogue9_f32(const GemmArgs& p, f32x16 (&acc)[2][2][4], int m_base, int n_base, int lane, char* wl) {
    if constexpr (!FULL)
        if (n_base >= p.N || m_base >= p.M) return;  // wave uniform
    float* out = reinterpret_cast<float*>(p.out);
    const int ldc = p.N;
    const int h = lane >> 5, lm = lane & 31;
    f32x4 old[D][4];
    auto load_old = [&](f32x4 (&dst)[4], int piece) __attribute__((always_inline)) {
        const int i = piece >> 2, jb = piece & 3;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pc = it * 64 + lane;
            const int m = FULL ? m_base + 32 * i + (pc >> 3) : min(m_base + 32 * i + (pc >> 3), p.M - 1);
            const int n = FULL ? n_base + 32 * jb + (pc & 7) * 4 : min(n_base + 32 * jb + (pc & 7) * 4, p.N - 4);
            dst[it] = *reinterpret_cast<const f32x4*>(out + (size_t)m * ldc + n);
        }
    };
    if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_old(old[d], d);
    }
#pragma unroll
    for (int piece = 0; piece < 16; ++piece) {
        const int i = piece >> 2, jb = piece & 3;
        char* sl = wl + (piece & 1) * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[jb >> 1][jb & 1][i][4 * g + e];
            if constexpr (EPI == EPI_GELU_F32) gelu_fast_x4(v);
            *reinterpret_cast<f32x4*>(sl + lm * 128 + (((2 * g + h) ^ (lm & 7)) << 4)) = f32x4{v[0], v[1], v[2], v[3]};
        }
        f32x4 vv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pc = it * 64 + lane;
            const int r = pc >> 3, cc = pc & 7;
            vv[it] = *reinterpret_cast<const f32x4*>(sl + r * 128 + ((cc ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pc = it * 64 + lane;
            const int r = pc >> 3, cc = pc & 7;
            f32x4 v = vv[it];
            if constexpr (EPI == EPI_RESID_F32) {
                const f32x4 o = old[piece % D][it];
                v = f32x4{o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]};
            }
            const int m = m_base + 32 * i + r, n = n_base + 32 * jb + cc * 4;
            if (FULL || (m < p.M && n < p.N)) *reinterpret_cast<f32x4*>(out + (size_t)m * ldc + n) = v;
        }
        if constexpr (EPI == EPI_RESID_F32)
            if (piece + D < 16) load_old(old[piece % D], piece + D);
    }
}

// --------------------------------------------------------------------------------------------
// kernel.  VAR bit 0: DMA schedule B (6 + 5 + 5 pieces over ks 3 / 0 / 1 instead of 8 + 8 over ks 3 / 0);
// bit 2 (4): REGISTER STAGING instead of LDS-DMA — the wave's 16 pieces of a stream position are fetched with
// global_load_dwordx4 into 64 VGPRs THREE positions ahead (two register sets = two positions in flight besides the
// two LDS buffers) and written to the LDS with ds_write_b128 one position ahead.  Measured motive
// (profiles/r3_gemm9_first_call.log): with LDS-DMA the loop is bound by (bytes in flight) / (loaded fabric latency) —
// the landing space is the LDS itself, so at most one K tile per wave can be in flight and every vmcnt wait sits
// behind an L2 miss; registers double the bytes in flight.  Needs an even number of K tiles (static set index).
// bits 4.. = DBG timing experiments (results are wrong): 16 no MFMA, 32 no staging loads, 64 no fragment reads,
// 128 no epilogue.
// --------------------------------------------------------------------------------------------
template <typename T, int EPI, int VAR = 0>
__global__ __launch_bounds__(256, 1) void gemm9_kernel(GemmArgs p, unsigned long long* timing) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Op<T>::v8;
    constexpr bool SCHED_B = (VAR & 1) != 0, REGST = (VAR & 4) != 0;
    constexpr bool NO_MFMA = (VAR & 16) != 0, NO_DMA = (VAR & 32) != 0, NO_RD = (VAR & 64) != 0, NO_EPI = (VAR & 128) != 0;
    constexpr int Q3 = SCHED_B ? 6 : 8;    // pieces issued in ks 3 (first part of a position)
    constexpr int Q0 = SCHED_B ? 11 : 16;  // ks 0 issues [Q3, Q0), ks 1 issues [Q0, 16)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nk = p.K >> 6;
    const unsigned rb = (unsigned)p.K * 2u;  // operand row stride in bytes

    // ---- static persistent schedule (gemm8's: XCD-contiguous ranges of a column-panel blocked tile order) ----
    const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8;
    const int total = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int n_my = (cnt > slot) ? (cnt - slot + nslot - 1) / nslot : 0;
    if (n_my == 0) return;
    const int panel_c = p.panel_c > 0 ? p.panel_c : tiles_n;
    const int panel_full = tiles_m * panel_c;
    auto tile_coords = [&](int it, int& tmi, int& tni) __attribute__((always_inline)) {
        const int o = start + slot + it * nslot;
        const int pnl = o / panel_full;
        const int rem = o - pnl * panel_full;
        const int w = min(panel_c, tiles_n - pnl * panel_c);
        tmi = __builtin_amdgcn_readfirstlane(rem / w);
        tni = __builtin_amdgcn_readfirstlane(pnl * panel_c + (rem - tmi * w));
    };

    // ---- LDS-DMA stream of this wave: 16 pieces of 8 rows per stream position -------------------------------
    // The stream state is scalar (operand panel + K offset, last valid row of the panel, position); a piece's
    // per-lane source offset is recomputed when it is issued (3 VALU instructions beside the MFMAs): lane l of
    // piece q fetches 16-byte chunk (l & 7) ^ swizzle(row) of row min(srow0 + 8 q + l / 8, lim).
    const bool is_a = wave < 2;
    const int srow0 = (wave & 1) * 128;  // first of the wave's 128 rows inside the tile's operand panel
    const char* s_base;                  // operand panel of the stream's tile + K offset (wave uniform)
    int s_kt, s_it, s_lim;
    const int rl = srow0 + (lane >> 3);
    // (row >> 1) & 7 of row = srow0 + 8 q + lane / 8:  4 (q & 1) + ((lane >> 4) & 3)
    const unsigned ch_even = (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) << 4);
    const unsigned ch_odd = (unsigned)(((lane & 7) ^ (4 + ((lane >> 4) & 3))) << 4);
    auto set_tile = [&](int it) __attribute__((always_inline)) {
        int tmi, tni;
        tile_coords(it, tmi, tni);
        s_it = it;
        s_kt = 0;
        s_lim = is_a ? p.M - tmi * 256 - 1 : p.N - tni * 256 - 1;
        s_base = is_a ? reinterpret_cast<const char*>(p.A) + (size_t)tmi * 256 * rb
                      : reinterpret_cast<const char*>(p.W) + (size_t)tni * 256 * rb;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        // past the end of the workgroup's tile list the last K tile is re-issued (into a dead buffer), so the
        // wait bookkeeping stays uniform
        if (s_kt + 1 < nk) {
            s_kt = s_kt + 1;
            s_base += 128;
        } else if (s_it + 1 < n_my) {
            set_tile(s_it + 1);
        }
    };
    // (loop-invariant rl + 8 q would be hoisted into 16 registers: the add stays beside its load)
    auto opaque = [](int v) __attribute__((always_inline)) {
        asm volatile("" : "+v"(v));
        return v;
    };
    auto issue1 = [&](int q, int buf) __attribute__((always_inline)) {  // q: compile-time constant after unrolling
        if constexpr (!NO_DMA) {
            const unsigned row = (unsigned)min(opaque(rl) + 8 * q, s_lim);
            const unsigned o = __umul24(row, rb) + ((q & 1) ? ch_odd : ch_even);  // row, rb < 2^24
            glds16(s_base + o, smem + buf * Q_BUF + wave * 16384 + q * 1024);
        }
    };
#define ESMK_ISSUE(QA, QB, BUF) \
    { _Pragma("unroll") for (int q_ = (QA); q_ < (QB); ++q_) issue1(q_, (BUF)); }
    // register staging (REGST): piece q of the stream's position -> G[set][q]; G[set][q] -> its LDS slot
    f32x4 G[2][16];
    auto gload = [&](int set, int q) __attribute__((always_inline)) {
        if constexpr (!NO_DMA) {
            const unsigned row = (unsigned)min(opaque(rl) + 8 * q, s_lim);
            const unsigned o = __umul24(row, rb) + ((q & 1) ? ch_odd : ch_even);
            G[set][q] = *reinterpret_cast<const f32x4*>(s_base + o);
        } else {
            G[set][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int lds_lane = wave * 16384 + lane * 16;
    auto lwrite = [&](int set, int q, int buf) __attribute__((always_inline)) {
        *reinterpret_cast<f32x4*>(smem + buf * Q_BUF + lds_lane + q * 1024) = G[set][q];
    };

    // ---- fragment reads ----------------------------------------------------------------------------------------
    const int lrow = (lane & 31) * 128;
    const int swz = (lane >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xo[ks] = ((2 * ks + (lane >> 5)) ^ swz) << 4;
    const int a_off = wr * 16384 + lrow;
    const int w_off = Q_WOFF + wc * 16384 + lrow;
    V8 fa[2][4], fw[2][4];
    if constexpr (NO_RD) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) fa[s][i][e] = fw[s][i][e] = Op<T>::from(0.f);
    }
    // one fragment of K sub-step ks from buffer `bp`: r even -> activation rows 32 (r/2) .., r odd -> weight rows 32 (r/2) ..
    auto rd1 = [&](V8 (&a)[4], V8 (&w)[4], const char* bp, int ks, int r) __attribute__((always_inline)) {
        if constexpr (!NO_RD) {
            if (r & 1) w[r >> 1] = *reinterpret_cast<const V8*>(bp + w_off + (r >> 1) * 4096 + xo[ks]);
            else a[r >> 1] = *reinterpret_cast<const V8*>(bp + a_off + (r >> 1) * 4096 + xo[ks]);
        }
    };

    f32x16 acc[2][2][4];  // [64-column half][32-column block][32-row block]
    // bias broadcast of 32-column block j (the C operand of a tile's first MFMAs: acc = bias + A.W^T, the order gemm8
    // uses).  Column n = n_base + 32 j + 8 (r >> 2) + 4 (lane >> 5) + (r & 3) sits in register r.  Scalar loads through
    // the constant address space: they do not enter the vmcnt queue.  EPI_V_T: the bias varies with the lane, its
    // epilogue adds it.
    typedef const __attribute__((address_space(4))) float* cfloat_ptr;
    int bias_n0 = 0;  // n_base of the current tile
    auto bias_vec = [&](int j) __attribute__((always_inline)) {
        f32x16 b;
        bool done = false;
        if constexpr (EPI != EPI_V_T) {
            if (p.bias != nullptr) {
                const int hsel = lane >> 5;
                if (bias_n0 + 128 <= p.N) {
                    cfloat_ptr cb = (cfloat_ptr)(unsigned long long)(p.bias + bias_n0 + 32 * j);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float b0 = cb[8 * (r >> 2) + (r & 3)], b1 = cb[8 * (r >> 2) + 4 + (r & 3)];
                        b[r] = hsel ? b1 : b0;
                    }
                } else {  // N tail: clamped vector loads
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = bias_n0 + 32 * j + 8 * (r >> 2) + 4 * hsel + (r & 3);
                        b[r] = n < p.N ? p.bias[n] : 0.f;
                    }
                }
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int r = 0; r < 16; ++r) b[r] = 0.f;
        }
        return b;
    };
    // MFMA t of a 16-wide K sub-step (16 MFMAs, every accumulator once: column block j = t / 4, row block i = t & 3).
    // first: the tile's first sub-step, C operand = bias broadcast bj.
    auto mma1 = [&](const V8 (&a)[4], const V8 (&w)[4], int t, bool first, const f32x16& bj) __attribute__((always_inline)) {
        const int j = t >> 2, i = t & 3;
        f32x16& c = acc[j >> 1][j & 1][i];
        if constexpr (NO_MFMA) {
            asm volatile("" ::"v"(a[i]), "v"(w[j]));
            if (first) c = bj;
        } else if constexpr (EPI == EPI_V_T) {  // lane owns 4 consecutive tokens of one channel
            c = first ? Op<T>::mma(a[i], w[j], bj) : Op<T>::mma(a[i], w[j], c);
        } else {  // lane owns 4 consecutive channels of one token
            c = first ? Op<T>::mma(w[j], a[i], bj) : Op<T>::mma(w[j], a[i], c);
        }
    };

    int cur = 0;
    // One 16-wide K sub-step as 16 pinned micro-steps of one MFMA: steps 0-7 also read one fragment of the NEXT
    // sub-step (>= 8 MFMA slots to land before that sub-step starts), steps 8-15 issue this sub-step's LDS-DMA
    // pieces [qa, qb) into buffer dbuf, evenly spread.  (hipcc's own interleave of the 16 + 8 + 8 instructions
    // bunched the reads and DMA pieces; sched_group_barrier did not separate the DMA instructions.)
    // REGST: steps 8-15 instead move pieces [qa, qb): G[gset][q] -> LDS buffer dbuf (position s+1), then refill
    // G[gset][q] from the stream (position s+3).
    auto substep = [&](const V8 (&a)[4], const V8 (&w)[4], V8 (&na)[4], V8 (&nw)[4], const char* nbuf, int nks,
                       int qa, int qb, int dbuf, bool first, int gset = 0) __attribute__((always_inline)) {
        f32x16 bj;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (first && (t & 3) == 0) bj = bias_vec(t >> 2);
            mma1(a, w, t, first, bj);
            if (t < 8) {
                rd1(na, nw, nbuf, nks, t);
            } else {
                const int cntq = qb - qa;
#pragma unroll
                for (int k = 0; k < cntq; ++k)
                    if (8 + (k * 8) / cntq == t) {
                        if constexpr (REGST) {
                            lwrite(gset, qa + k, dbuf);
                            gload(gset, qa + k);
                        } else {
                            issue1(qa + k, dbuf);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // par: parity of the stream position (REGST: position s moves register set par ^ 1)
    auto ktile = [&](bool first, int par) __attribute__((always_inline)) {
        const char* sb = smem + cur * Q_BUF;
        const char* sn = smem + (cur ^ 1) * Q_BUF;
        if constexpr (REGST) {
            // position s+1: registers -> buffer cur^1 (free since the barrier of position s-1); refill with s+3
            substep(fa[0], fw[0], fa[1], fw[1], sb, 1, 0, 6, cur ^ 1, first, par ^ 1);    // ks 0
            substep(fa[1], fw[1], fa[0], fw[0], sb, 2, 6, 11, cur ^ 1, false, par ^ 1);   // ks 1
            substep(fa[0], fw[0], fa[1], fw[1], sb, 3, 11, 16, cur ^ 1, false, par ^ 1);  // ks 2
            // every wave's LDS writes of position s+1 are done, every wave is done reading buffer cur; the global
            // loads stay in flight (hipcc counts them: a ds_write waits for exactly its own load)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            advance();
            __builtin_amdgcn_sched_barrier(0);
            substep(fa[1], fw[1], fa[0], fw[0], sn, 0, 0, 0, cur, false);                 // ks 3
        } else {
            substep(fa[0], fw[0], fa[1], fw[1], sb, 1, Q3, Q0, cur ^ 1, first);   // ks 0
            substep(fa[1], fw[1], fa[0], fw[0], sb, 2, Q0, 16, cur ^ 1, false);   // ks 1
            substep(fa[0], fw[0], fa[1], fw[1], sb, 3, 0, 0, cur, false);         // ks 2
            // every wave's pieces of position s+1 have landed, every wave is done reading buffer cur
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            advance();
            __builtin_amdgcn_sched_barrier(0);
            substep(fa[1], fw[1], fa[0], fw[0], sn, 0, 0, Q3, cur, false);        // ks 3
        }
        cur ^= 1;
    };

    // ---- prologue: position 0 completely, the first part of position 1 ----------------------------------------
    set_tile(0);
    if constexpr (REGST) {
        // positions 0, 1 -> G[0], G[1]; position 0 -> LDS buffer 0; position 2 -> G[0]; the stream stands at position 3
#pragma unroll
        for (int q = 0; q < 16; ++q) gload(0, q);
        advance();
#pragma unroll
        for (int q = 0; q < 16; ++q) gload(1, q);
        advance();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            lwrite(0, q, 0);
            gload(0, q);
        }
        advance();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        ESMK_ISSUE(0, 16, 0)
        advance();
        ESMK_ISSUE(0, Q3, 1)
        if constexpr (SCHED_B) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) rd1(fa[0], fw[0], smem, 0, r);

    auto stamp = [&](int it, int k) __attribute__((always_inline)) {
        if (timing != nullptr && tid == 0) {
            unsigned long long* t = timing + ((size_t)blockIdx.x * 32 + (it & 31)) * 4;
            t[k] = __builtin_readcyclecounter();
            if (k == 2) t[3] = wall_clock64();
        }
    };

    for (int it = 0; it < n_my; ++it) {
        int tmi, tni;
        tile_coords(it, tmi, tni);
        const int m_base = tmi * 256 + wr * 128, n_base = tni * 256 + wc * 128;
        bias_n0 = n_base;
        stamp(it, 0);
        if constexpr (REGST) {  // nk is even: static register-set index
            ktile(true, 0);
            ktile(false, 1);
#pragma unroll 1
            for (int kt = 2; kt < nk; kt += 2) {
                ktile(false, 0);
                ktile(false, 1);
            }
        } else {
            ktile(true, 0);
#pragma unroll 1
            for (int kt = 1; kt < nk; ++kt) ktile(false, 0);
        }
        stamp(it, 1);
        char* slice = smem + Q_EPI + wave * Q_SLICE;
        if constexpr (NO_EPI) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(acc[hf][j][i]));
        } else if constexpr (EPI == EPI_STORE_F32 || EPI == EPI_GELU_F32 || EPI == EPI_RESID_F32) {
            const bool full = (m_base + 128 <= p.M) && (n_base + 128 <= p.N);
            constexpr int D = REGST ? 3 : 8;  // REGST keeps 128 staging registers live across the epilogue
            if (full) epilogue9_f32<T, EPI, true, D>(p, acc, m_base, n_base, lane, slice);
            else epilogue9_f32<T, EPI, false, D>(p, acc, m_base, n_base, lane, slice);
        } else {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int nb = n_base + 64 * hf;
                bool full = (m_base + 128 <= p.M) && (nb + 64 <= p.N);
                if constexpr (EPI == EPI_V_T) full = full && (p.T % 32 == 0);
                if (full) epilogue8<T, EPI, true, false, false, 4>(p, acc[hf], m_base, nb, lane, slice + hf * 4096, 0, 0, 0);
                else epilogue8<T, EPI, false, false, false, 4>(p, acc[hf], m_base, nb, lane, slice + hf * 4096, 0, 0, 0);
            }
        }
        stamp(it, 2);
    }
    wait_vmcnt0();  // the trailing (dummy) DMA writes must land before the LDS is released
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
static unsigned long long* g_timing9 = nullptr;
void gemm9_set_timing(unsigned long long* dev_buf) { g_timing9 = dev_buf; }

static int num_workgroups9() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        n = cus / 8 * 8;
    }
    return n;
}

template <typename T, int EPI, int VAR = 0>
static hipError_t launch9(GemmArgs p, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm9_kernel<T, EPI, VAR>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_n = (p.N + 255) / 256;
    if (p.panel_c <= 0) {  // gemm8's choice: the 32 concurrent tiles of an XCD form a block as square as possible
        if (tiles_n <= 6) p.panel_c = tiles_n;
        else if (tiles_n % 5 == 0) p.panel_c = 5;
        else if (tiles_n % 4 == 0) p.panel_c = 4;
        else if (tiles_n % 6 == 0) p.panel_c = 6;
        else p.panel_c = 5;
    }
    hipLaunchKernelGGL(kern, dim3(num_workgroups9()), dim3(256), Q_LDS, st, p, g_timing9);
    return hipGetLastError();
}

bool gemm9_supports(const GemmArgs& p, int epi) {
    if (p.K % 64 != 0 || p.N % 8 != 0 || p.M <= 0) return false;
    if (gemm8_generalised(p, epi) || p.half_m > 0) return false;
    if ((epi == EPI_QKV_ROPE || epi == EPI_V_T) && p.N % 64 != 0) return false;
    return epi >= EPI_STORE_T && epi <= EPI_V_T;
}

template <typename T>
static hipError_t dispatch9(const GemmArgs& p, int epi, int var, hipStream_t st) {
#define ESMK_CASES9(V)                                                      \
    switch (epi) {                                                          \
        case EPI_STORE_T: return launch9<T, EPI_STORE_T, V>(p, st);         \
        case EPI_STORE_F32: return launch9<T, EPI_STORE_F32, V>(p, st);     \
        case EPI_GELU_T: return launch9<T, EPI_GELU_T, V>(p, st);           \
        case EPI_GELU_F32: return launch9<T, EPI_GELU_F32, V>(p, st);       \
        case EPI_RESID_F32: return launch9<T, EPI_RESID_F32, V>(p, st);     \
        case EPI_QKV_ROPE: return launch9<T, EPI_QKV_ROPE, V>(p, st);       \
        case EPI_V_T: return launch9<T, EPI_V_T, V>(p, st);                 \
    }
    if (var == 0) { ESMK_CASES9(0) }
    if (var == 1) { ESMK_CASES9(1) }
    if (var == 4) {
        if ((p.K / 64) % 2 != 0) return hipErrorInvalidValue;  // register staging: even number of K tiles
        ESMK_CASES9(4)
    }
    if constexpr (std::is_same<T, _Float16>::value) {
        // measurement variants (tools/bench_gemm9.py): plain-store / residual epilogues only
        if (epi == EPI_STORE_T) {
            switch (var) {
                case 36: return launch9<T, EPI_STORE_T, 36>(p, st);  // register staging without its loads
                case 16: return launch9<T, EPI_STORE_T, 16>(p, st);
                case 32: return launch9<T, EPI_STORE_T, 32>(p, st);
                case 64: return launch9<T, EPI_STORE_T, 64>(p, st);
                case 128: return launch9<T, EPI_STORE_T, 128>(p, st);
                case 96: return launch9<T, EPI_STORE_T, 96>(p, st);
                case 224: return launch9<T, EPI_STORE_T, 224>(p, st);
            }
        }
    }
#undef ESMK_CASES9
    return hipErrorInvalidValue;
}

hipError_t launch_gemm9(const GemmArgs& p, int epi, int operand_dtype, int var, hipStream_t st) {
    if (!gemm9_supports(p, epi)) return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_F16) return dispatch9<_Float16>(p, epi, var, st);
    if (operand_dtype == ESMK_DT_BF16) return dispatch9<__bf16>(p, epi, var, st);
    return hipErrorInvalidValue;
}

}  // namespace esmk
