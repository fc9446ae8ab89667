// gemm8_v1.hip — frozen copy of the first persistent GEMM (commit d2404fb), kept ONLY as an in-run
// A/B reference for tools/microbench.py (box-to-box clock variance is larger than the deltas under study).
// Not used by the engine.
// gemm8.hip — persistent, ping-pong scheduled nn.Linear for gfx950:
//     C[M,N] = A[M,K] . W[N,K]^T (+ bias, + fused epilogue)          K % 64 == 0, N % 8 == 0
//
// Replaces the same reference calls as gemm.hip (esm/multihead_attention.py:256-261,395;
// esm/modules.py:138-139,309,313) and carries the same fused epilogues; it is the kernel the
// engine uses for every large projection of the layer stack.
//
// Structure (one workgroup per CU, 8 waves, 256x256 output tile, K step 64):
//   * PERSISTENT: 256 workgroups walk a static tile list.  Workgroup b runs on XCD b % 8; each XCD
//     owns a contiguous range of the tile order and its 32 workgroups always work on 32 consecutive
//     tiles of that order.  The order is "column-panel blocked" (panels of `panel_c` N tiles, M
//     fastest across the panel rows), so the 32 concurrent tiles of an XCD form a ~(32/C) x C block
//     and share their A / W K-slabs through the XCD's private L2.
//   * The K tiles of all tiles of a workgroup form ONE stream: the LDS-DMA (global_load_lds_dwordx4)
//     prefetch of stream position s+1 / s+2 runs during position s also across tile boundaries, so
//     a tile's first operands are already in the LDS when the previous tile's epilogue finishes.
//   * Every K tile is staged as four 16 KiB units U0 = A rows of the (i0,i1) fragments of both
//     wave groups, U1 = W rows of the j0 fragments of all four wave columns, U2 = W rows j1,
//     U3 = A rows (i2,i3).  Two K-tile buffers (2 x 64 KiB) + 8 x 4 KiB epilogue slices = 160 KiB.
//     LDS rows are 128 B, the 16-byte chunk index is XOR-swizzled with (row>>1)&7 on the DMA source
//     address and on the ds_read_b128 address (conflict free, see gemm.hip).
//   * PING-PONG: the two waves of a SIMD (wave w of group 0 = rows 0..127, wave w+4 of group 1 =
//     rows 128..255) run the same instruction stream shifted by one barrier.  A K tile is four
//     phases; each phase is a load section (ds_reads of the next fragments + 2 LDS-DMA
//     instructions + a COUNTED s_waitcnt vmcnt(8)) and a matrix section (8 v_mfma_f32_32x32x16 =
//     one 64x32 quadrant of the wave's 128x64 block over K = 64), separated by raw s_barrier.
//     While one group issues MFMAs the other one issues its LDS / DMA traffic.
//   * vmcnt is never drained in the loop: a unit is waited for 4 load sections (= 8 DMA
//     instructions per wave) after it was issued and read one section after that wait.
//
// Hazard bookkeeping (b_k = k-th barrier of a K tile in group-0 numbering; group 1 lags by one):
//   section   reads (buffer cur)        DMA issue                       wait
//   L0        U0 (A i0,i1), U1 (W j0)   U2 of position s+1 -> cur^1     vmcnt(8)  (U2 of cur landed)
//   L1        U2 (W j1)                 U3 of position s+1 -> cur^1     vmcnt(8)  (U3 of cur landed)
//   L2        U3 (A i2,i3)              U0 of position s+2 -> cur       -
//   L3        -                         U1 of position s+2 -> cur       vmcnt(8)  (U0,U1 of cur^1)
//   WAR: U0/U1 of `cur` are last read in L0 (retired before b_3 by both groups) and re-staged in
//   L2/L3 (after b_4); U2/U3 of cur^1 were last read one K tile earlier.  RAW: every unit is
//   waited for by ALL waves before a barrier that precedes its first read.
#include "common.h"
#include "kernels.h"

namespace esmk {
namespace v1ref {

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
ESMK_DEV void wait_vmcnt8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

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
// epilogue: the wave's 128 (m) x 64 (n) block leaves through a private 4 KiB LDS slice in
// 32-row pieces, so that every global store instruction covers whole 128-byte row segments.
//   normal orientation (MFMA A operand = weight rows):
//     acc[j][i][r]:  m = m_base + 32 i + (lane & 31);  n = n_base + 32 j + 8 (r>>2) + 4 (lane>>5) + (r&3)
//   EPI_V_T (MFMA A operand = activation rows):
//     acc[j][i][r]:  n = n_base + 32 j + (lane & 31);  m = m_base + 32 i + 8 (r>>2) + 4 (lane>>5) + (r&3)
// --------------------------------------------------------------------------------------------
template <typename T, int EPI>
ESMK_DEV void epilogue8(const GemmArgs& p, f32x16 (&acc)[2][4], int m_base, int n_base, int lane,
                        char* wl) {
    using V4 = typename Op<T>::v4;
    using V8 = typename Op<T>::v8;
    const int h = lane >> 5, lm = lane & 31;
    if (n_base >= p.N || m_base >= p.M) return;  // wave uniform

    if constexpr (EPI == EPI_V_T) {
        // vt[b][head][dv][Tp], keys permuted inside groups of 16 (4-groups 1 and 2 swapped)
        const int head = n_base >> 6;
        T* vt = reinterpret_cast<T*>(p.vt);
        const float bv0 = p.bias[n_base + lm], bv1 = p.bias[n_base + 32 + lm];
        const bool aligned = (p.T % 32 == 0);  // a 32-token piece = one aligned run of one sequence
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // LDS piece: 64 rows (dv) x 64 B (32 tokens); 16-byte chunk c holds 8 token slots
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int dv = 32 * j + lm;
                const float bv = j ? bv1 : bv0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // tokens 8g + 4h + e (e = 0..3) of the piece
                    const int chunk = aligned ? 2 * (g >> 1) + h : g;
                    const int half = aligned ? (g & 1) : h;
                    *reinterpret_cast<V4*>(wl + dv * 64 + ((chunk ^ ((dv >> 1) & 3)) << 4) + 8 * half) =
                        pack4_<T>(acc[j][i][4 * g] + bv, acc[j][i][4 * g + 1] + bv,
                                  acc[j][i][4 * g + 2] + bv, acc[j][i][4 * g + 3] + bv);
                }
            }
            const int mp = m_base + 32 * i;
            if (aligned) {
                if (mp < p.M) {
                    const int b = mp / p.T, t0 = mp - b * p.T;
                    T* base = vt + ((size_t)(b * p.H + head) * 64) * p.Tp + t0;
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
                        *reinterpret_cast<V8*>(base + (size_t)r * p.Tp + c * 8) = v[it];
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
                    const int tp = (t & ~15) | ((((t16 >> 2) & 1) << 3) | (((t16 >> 3) & 1) << 2) | (t16 & 3));
                    vt[((size_t)(b * p.H + head) * 64 + r) * (size_t)p.Tp + tp] = v;
                }
            }
        }
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        // q and k projections (N = 2E): the wave's 64 columns are exactly one head (head_dim 64)
        const int which = n_base / p.E;  // 0 q, 1 k (wave uniform)
        const int head = (n_base - which * p.E) >> 6;
        T* qk = reinterpret_cast<T*>(which == 0 ? p.q : p.k);
        const float sc = which == 0 ? p.scaling : 1.0f;
        f32x4 b1[4], b2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            b1[g] = *reinterpret_cast<const f32x4*>(p.bias + n_base + 8 * g + 4 * h);
            b2[g] = *reinterpret_cast<const f32x4*>(p.bias + n_base + 32 + 8 * g + 4 * h);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m_base + 32 * i + lm, p.M - 1);
            const int t = m % p.T;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 8 * g + 4 * h;  // first of 4 consecutive dims in [0,32)
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos + (size_t)t * 32 + d0);
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.sin + (size_t)t * 32 + d0);
                float y1[4], y2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // bias, q scaling (mha.py:261), x*cos + rotate_half(x)*sin (rotary_embedding.py:11-20)
                    const float a1 = (acc[0][i][4 * g + e] + b1[g][e]) * sc;
                    const float a2 = (acc[1][i][4 * g + e] + b2[g][e]) * sc;
                    y1[e] = a1 * c[e] - a2 * s[e];
                    y2[e] = a2 * c[e] + a1 * s[e];
                }
                *reinterpret_cast<V4*>(wl + lm * 128 + ((g ^ (lm & 7)) << 4) + 8 * h) =
                    pack4_<T>(y1[0], y1[1], y1[2], y1[3]);
                *reinterpret_cast<V4*>(wl + lm * 128 + (((4 + g) ^ (lm & 7)) << 4) + 8 * h) =
                    pack4_<T>(y2[0], y2[1], y2[2], y2[3]);
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
                if (mm < p.M) {
                    const int b = mm / p.T, tt = mm - b * p.T;
                    *reinterpret_cast<V8*>(qk + ((size_t)(b * p.H + head) * p.T + tt) * 64 + cc * 8) = v[it];
                }
            }
        }
    } else if constexpr (EPI == EPI_STORE_T || EPI == EPI_GELU_T) {
        T* out = reinterpret_cast<T*>(p.out);
        f32x4 bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n_base + 32 * j + 8 * g + 4 * h;
                bv[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bv[j][g] = *reinterpret_cast<const f32x4*>(p.bias + n);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[j][i][4 * g + e] + bv[j][g][e];
                        if constexpr (EPI == EPI_GELU_T) v[e] = gelu_fast(v[e]);
                    }
                    *reinterpret_cast<V4*>(wl + lm * 128 + (((4 * j + g) ^ (lm & 7)) << 4) + 8 * h) =
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
                if (m < p.M && n < p.N) *reinterpret_cast<V8*>(out + (size_t)m * p.N + n) = v[it];
            }
        }
    } else {
        // fp32 outputs: 8 pieces of 32 rows x 32 columns (128-byte row segments)
        float* out = reinterpret_cast<float*>(p.out);
        f32x4 old[4], nxt[4];
        auto load_old = [&](f32x4 (&dst)[4], int piece) {
            const int i = piece >> 1, j = piece & 1;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pc = it * 64 + lane;
                const int m = min(m_base + 32 * i + (pc >> 3), p.M - 1);
                const int n = min(n_base + 32 * j + (pc & 7) * 4, p.N - 4);
                dst[it] = *reinterpret_cast<const f32x4*>(out + (size_t)m * p.N + n);
            }
        };
        if constexpr (EPI == EPI_RESID_F32) load_old(old, 0);
#pragma unroll
        for (int piece = 0; piece < 8; ++piece) {
            const int i = piece >> 1, j = piece & 1;
            if constexpr (EPI == EPI_RESID_F32)
                if (piece + 1 < 8) load_old(nxt, piece + 1);  // residual of the next piece in flight
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n_base + 32 * j + 8 * g + 4 * h;
                f32x4 bvv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bvv = *reinterpret_cast<const f32x4*>(p.bias + n);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[j][i][4 * g + e] + bvv[e];
                    if constexpr (EPI == EPI_GELU_F32) v[e] = gelu_fast(v[e]);
                }
                *reinterpret_cast<f32x4*>(wl + lm * 128 + (((2 * g + h) ^ (lm & 7)) << 4)) = v;
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
                if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(out + (size_t)m * p.N + n) = v;
            }
            if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
                for (int it = 0; it < 4; ++it) old[it] = nxt[it];
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// kernel
// --------------------------------------------------------------------------------------------
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Op<T>::v8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // M half of the tile AND ping-pong group
    const int wn = wave & 3;    // 64-column slice of the tile
    const int nk = p.K >> 6;
    const unsigned row_bytes = (unsigned)p.K * 2u;

    // ---- static persistent schedule -----------------------------------------------------------
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
    auto tile_coords = [&](int it, int& tmi, int& tni) {
        const int o = start + slot + it * nslot;
        const int pnl = o / panel_full;
        const int rem = o - pnl * panel_full;
        const int w = min(panel_c, tiles_n - pnl * panel_c);
        tmi = rem / w;
        tni = pnl * panel_c + (rem - tmi * w);
    };

    // ---- LDS-DMA streams -----------------------------------------------------------------------
    // unit row handled by this lane in DMA instruction q (0,1) of a unit: ur = 16 wave + 8 q + lane/8
    const int ur0 = 16 * wave + (lane >> 3), ur1 = ur0 + 8;
    const unsigned cs0 = (unsigned)(((lane & 7) ^ ((ur0 >> 1) & 7)) << 4);
    const unsigned cs1 = (unsigned)(((lane & 7) ^ ((ur1 >> 1) & 7)) << 4);
    struct Stream {
        const char* base;    // operand panel of the stream's tile + K offset (wave uniform)
        int kt, it;          // K tile inside the tile, tile iteration
        unsigned off0, off1; // per-lane byte offsets of the two DMA instructions
    };
    auto set_tile = [&](Stream& s, int unit, int it) {
        int tmi, tni;
        tile_coords(it, tmi, tni);
        s.it = it;
        s.kt = 0;
        if (unit == 0 || unit == 3) {  // A rows: group (ur>>6) * 128 + (unit 3 ? 64 : 0) + (ur & 63)
            const int lim = p.M - tmi * 256 - 1;
            const int add = unit == 3 ? 64 : 0;
            s.base = reinterpret_cast<const char*>(p.A) + (size_t)tmi * 256 * row_bytes;
            s.off0 = (unsigned)min((ur0 >> 6) * 128 + add + (ur0 & 63), lim) * row_bytes + cs0;
            s.off1 = (unsigned)min((ur1 >> 6) * 128 + add + (ur1 & 63), lim) * row_bytes + cs1;
        } else {  // W rows: wave column (ur>>5) * 64 + (unit 2 ? 32 : 0) + (ur & 31)
            const int lim = p.N - tni * 256 - 1;
            const int add = unit == 2 ? 32 : 0;
            s.base = reinterpret_cast<const char*>(p.W) + (size_t)tni * 256 * row_bytes;
            s.off0 = (unsigned)min((ur0 >> 5) * 64 + add + (ur0 & 31), lim) * row_bytes + cs0;
            s.off1 = (unsigned)min((ur1 >> 5) * 64 + add + (ur1 & 31), lim) * row_bytes + cs1;
        }
    };
    auto issue = [&](Stream& s, int unit, int buf) {
        char* dst = smem + buf * P_BUF + unit * P_UNIT + wave * 2048;
        glds16(s.base + s.off0, dst);
        glds16(s.base + s.off1, dst + 1024);
        // advance to the next stream position; past the end of the workgroup's tile list the last
        // K tile is re-issued (into a dead buffer) so the vmcnt bookkeeping stays uniform
        if (s.kt + 1 < nk) {
            ++s.kt;
            s.base += 128;
        } else if (s.it + 1 < n_my) {
            set_tile(s, unit, s.it + 1);
        }
    };

    Stream sA0, sW0, sW1, sA1;  // U0, U1, U2, U3
    set_tile(sA0, 0, 0);
    set_tile(sW0, 1, 0);
    set_tile(sW1, 2, 0);
    set_tile(sA1, 3, 0);
    issue(sA0, 0, 0);
    issue(sW0, 1, 0);
    issue(sW1, 2, 0);
    issue(sA1, 3, 0);
    issue(sA0, 0, 1);
    issue(sW0, 1, 1);

    // ---- fragment read offsets -----------------------------------------------------------------
    const int lrow = (lane & 31) * 128;
    const int swz = (lane >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xo[ks] = ((2 * ks + (lane >> 5)) ^ swz) << 4;
    const int a_off = grp * (64 * 128) + lrow;
    const int w_off = wn * (32 * 128) + lrow;

    V8 fa[2][4], fw0[4], fw1[4];
    f32x16 acc[2][4];

    auto rdA = [&](const char* ub) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[i2][ks] = *reinterpret_cast<const V8*>(ub + a_off + i2 * 4096 + xo[ks]);
    };
    auto rdW = [&](V8 (&fw)[4], const char* ub) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fw[ks] = *reinterpret_cast<const V8*>(ub + w_off + xo[ks]);
    };
    // one 64 x 32 quadrant over K = 64: 8 MFMAs, the two accumulators alternate
    auto quad = [&](f32x16& c0, f32x16& c1, const V8 (&fw)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (EPI == EPI_V_T) {  // lane owns 4 consecutive tokens of one channel
                c0 = Op<T>::mma(fa[0][ks], fw[ks], c0);
                c1 = Op<T>::mma(fa[1][ks], fw[ks], c1);
            } else {  // lane owns 4 consecutive channels of one token
                c0 = Op<T>::mma(fw[ks], fa[0][ks], c0);
                c1 = Op<T>::mma(fw[ks], fa[1][ks], c1);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    wait_vmcnt8();  // U0, U1 of position 0 have landed
    wg_barrier();

    int cur = 0;
    for (int it = 0; it < n_my; ++it) {
        int tmi, tni;
        tile_coords(it, tmi, tni);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

        if (grp == 1) wg_barrier();  // group 1 runs one section behind group 0
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const char* sb = smem + cur * P_BUF;
            // ---- phase 0 -----------------------------------------------------------------------
            rdA(sb);
            rdW(fw0, sb + P_UNIT);
            issue(sW1, 2, cur ^ 1);
            wait_vmcnt8();
            wg_barrier();
            quad(acc[0][0], acc[0][1], fw0);
            wg_barrier();
            // ---- phase 1 -----------------------------------------------------------------------
            rdW(fw1, sb + 2 * P_UNIT);
            issue(sA1, 3, cur ^ 1);
            wait_vmcnt8();
            wg_barrier();
            quad(acc[1][0], acc[1][1], fw1);
            wg_barrier();
            // ---- phase 2 -----------------------------------------------------------------------
            rdA(sb + 3 * P_UNIT);
            issue(sA0, 0, cur);
            wg_barrier();
            quad(acc[1][2], acc[1][3], fw1);
            wg_barrier();
            // ---- phase 3 -----------------------------------------------------------------------
            issue(sW0, 1, cur);
            wait_vmcnt8();
            wg_barrier();
            quad(acc[0][2], acc[0][3], fw0);
            wg_barrier();
            cur ^= 1;
        }
        if (grp == 0) wg_barrier();  // re-align: both groups run the epilogue together

        epilogue8<T, EPI>(p, acc, tmi * 256 + grp * 128, tni * 256 + wn * 64, lane,
                          smem + P_EPI + wave * P_SLICE);
    }
    wait_vmcnt0();  // the trailing (dummy) DMA writes must land before the LDS is released
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
static int num_workgroups() {
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

template <typename T, int EPI>
static hipError_t launch8(GemmArgs p, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm8_kernel<T, EPI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_n = (p.N + 255) / 256;
    if (p.panel_c <= 0) {
        // 32 concurrent tiles per XCD should form a block as square as possible: ~6 x 5 or 8 x 4
        if (tiles_n <= 6) p.panel_c = tiles_n;
        else if (tiles_n % 5 == 0) p.panel_c = 5;
        else if (tiles_n % 4 == 0) p.panel_c = 4;
        else if (tiles_n % 6 == 0) p.panel_c = 6;
        else p.panel_c = 5;
    }
    hipLaunchKernelGGL(kern, dim3(num_workgroups()), dim3(512), P_LDS, st, p);
    return hipGetLastError();
}

template <typename T>
static hipError_t dispatch8(const GemmArgs& p, int epi, hipStream_t st) {
    switch (epi) {
        case EPI_STORE_T: return launch8<T, EPI_STORE_T>(p, st);
        case EPI_STORE_F32: return launch8<T, EPI_STORE_F32>(p, st);
        case EPI_GELU_T: return launch8<T, EPI_GELU_T>(p, st);
        case EPI_GELU_F32: return launch8<T, EPI_GELU_F32>(p, st);
        case EPI_RESID_F32: return launch8<T, EPI_RESID_F32>(p, st);
        case EPI_QKV_ROPE: return launch8<T, EPI_QKV_ROPE>(p, st);
        case EPI_V_T: return launch8<T, EPI_V_T>(p, st);
    }
    return hipErrorInvalidValue;
}

bool gemm8_supports_v1(const GemmArgs& p, int epi) {
    if (p.K % 64 != 0 || p.N % 8 != 0 || p.M <= 0) return false;
    if ((epi == EPI_QKV_ROPE || epi == EPI_V_T) && p.N % 64 != 0) return false;
    return true;
}

hipError_t launch_gemm8_v1(const GemmArgs& p, int epi, int operand_dtype, hipStream_t st) {
    if (!gemm8_supports_v1(p, epi)) return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_F16) return dispatch8<_Float16>(p, epi, st);
    if (operand_dtype == ESMK_DT_BF16) return dispatch8<__bf16>(p, epi, st);
    return hipErrorInvalidValue;
}

}  // namespace v1ref
}  // namespace esmk
