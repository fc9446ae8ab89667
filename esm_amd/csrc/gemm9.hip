// gemm9.hip — persistent nn.Linear for gfx950 with ONE wave per SIMD:
//     C[M,N] = A[M,K] . W[N,K]^T (+ bias, + fused epilogue)          K % 64 == 0, N % 8 == 0, dense operands
//
// Same contract, tile order and results as gemm8.hip (reference esm/multihead_attention.py:256-261,395;
// esm/modules.py:138-139), different main loop; since round 3 the kernel of every dense launch of the layer stack
// (gemm8 keeps the generalised-addressing calls: MSA Transformer, split weights, packed batches).  gemm8 runs 8 waves
// (two per SIMD) on 128 x 64 wave blocks and hides LDS / DMA latency by ping-pong between the two waves of a SIMD:
// 8 barriers per K tile, 24 KiB of fragment reads per 256 cycles of MFMA.  gemm9 runs 4 waves (one per SIMD, 256
// accumulator + 256 vector registers each) on 128 x 128 wave blocks of the same 256 x 256 x 64 tile:
//   * fragment reads per MFMA drop by a third (32 KiB per 2048 cycles of MFMA per wave);
//   * the wave's own instruction stream interleaves its 128 v_mfma_f32_16x16x32 (64 slots of two) with 32 ds_read_b128
//     and 16 LDS-DMA pieces, pinned with sched_barrier (hipcc's own interleave bunches them);
//   * TWO barriers per K tile, the LDS-DMA queue is never drained (counted vmcnt), a piece has > 1 K tile to land.
// Every output element sees the same MFMA sequence over K as in gemm8 (bias enters as the C operand of the first
// MFMA, then K ascending), so results are bit-identical to gemm8's (tools/bench_gemm9.py checks it on ragged shapes).
//
// What was measured on the way (profiles/r3_gemm9_*.log, M = 65536):
//   * v1: one barrier per K tile and `vmcnt(0)` before it: 3300 - 3500 cycles per K tile against gemm8's 2650 although
//     its fragment reads cost nothing (2200 cycles without the DMA) — a queue drained once per K tile idles half of the
//     time.  Staging through registers (global_load -> VGPRs -> ds_write_b128) did not help.
//   * schedule C (the vendor asm kernel's idea, hipBLASLt MT256x256x64_MI16x16x1, read with llvm-objdump): the
//     fragments of a WHOLE K tile live in registers (128 VGPRs) and are read early, so that the LDS buffer of K tile s
//     is free again a quarter into K tile s and the DMA of K tile s+2 streams into it: ties gemm8 (both on the 1400 W
//     cap at 1.6 GHz).
//   * 16x16x32 MFMAs instead of 32x32x16 (tools/mfma_power_probe.hip: 13 % less energy per flop — the shape the vendor
//     kernel uses): the clock rises to 1.8 - 1.9 GHz and the loop becomes DMA bound.
//   * a piece that finds the CU's memory pipeline busy stalls the wave — and with one wave per SIMD its MFMA issue;
//     4 waves x 1 piece per 2 slots is the pipeline's peak rate.  One piece per 3 slots: +8 .. 9 %.
//   * non-temporal output stores (the output tile is not read again by the launch and was pushing operand panels out
//     of the XCD's L2): +4 .. 7 % on the K = 1280 shapes.  Together: 1140 - 1270 TFLOP/s on the four layer shapes,
//     the hipBLASLt kernel's rate (same box: 1194 - 1284), gemm8 1060 - 1125.
//   * K offset staggered by tile column (the vendor's StaggerU): the DMA stream alone +12 %, the kernel +-0; operand
//     loads with nt / sc bits: worse (tools/dma_probe.hip: the stream alone runs 22.5 TB/s, L2-resident data 32 TB/s).
//
// LDS (160 KiB): two K-tile buffers of 64 KiB (A rows 0..255, then W rows 0..255; 128-byte rows, 16-byte chunk index
// XOR-swizzled with (row >> 1) & 7 on the DMA source address and on the ds_read_b128), then 4 x 8 KiB wave-private
// epilogue slices.  Wave w stages rows [64 w, 64 w + 64) of the A panel and of the W panel (8 + 8 pieces of 8 rows,
// `buffer_load ... lds`: rows past the end of the operand are out of the descriptor's range and arrive as zeros);
// wave (wr, wc) = (w >> 1, w & 1) computes rows [128 wr, +128) x columns [128 wc, +128) of the tile.
//
// One K tile (stream position s, LDS buffer cur = s & 1), MFMA slots m = 0..63 (two MFMAs each: K half m >> 5, blocks
// 2 (m & 31), 2 (m & 31) + 1 of the half's 8 x 8; X = fragments of K 0..31, Y = fragments of K 32..63 of the tile):
//     m  0..15   ds_read: Y fragments of position s (buffer cur) — X of position s was read during position s-1
//     m 16       s_waitcnt lgkmcnt(0); s_barrier        every wave has read buffer cur completely
//     m 16..61   LDS-DMA: the 16 pieces of position s+2 -> buffer cur, one per three slots
//     m 52       s_waitcnt vmcnt(12); s_barrier         every wave's pieces of position s+1 have landed (the 12
//                                                       youngest pieces, of position s+2, stay in flight)
//     m 52..59   ds_read: X fragments of position s+1 (buffer cur^1), two per slot
// The K tiles of all tiles of a workgroup form one stream, as in gemm8: the first operands of the next tile land
// during the epilogue.  Half-height tiles (small batches): see the HM parameter of the kernel.
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

#define ESMK_INL __attribute__((always_inline))

// --------------------------------------------------------------------------------------------
// Epilogues of the wave's 128 x 128 block.  The accumulators live in AGPRs; ds_write_b128 takes them from there, so
// every epilogue writes fp32 quads straight into the wave's LDS slice and does its arithmetic on the transposed
// (row-major) read — no v_accvgpr_read pass, few live VGPRs.
// acc[nj][mi][r] (16 x 16 x 32 MFMA blocks):  m = m_base + 16 mi + (lane & 15);  n = n_base + 16 nj + 4 (lane >> 4) + r
// --------------------------------------------------------------------------------------------
// fp32 outputs: 16 pieces of 32 rows x 32 columns (128-byte row segments).  EPI_RESID_F32 keeps D pieces of the
// residual tile in flight.  Same arithmetic as epilogue8 (old + value).
// (Round 5: the residual added in the L2 instead — one global_atomic_add_f32 per element, no return value, whole 128-byte
// row segments per request; the same single IEEE addition, bit-identical on the GPU — costs the out-projection 11.7 instead
// of 8.4 ms per step and fc2 24.6 instead of 23.0: the L2's atomic units do not sustain 84 M adds per launch.  Removed;
// profiles/r5_attention_w64_and_resid_atomic_ab.log.)
// NMI = 16-row blocks of the wave's block: 8 (128 rows), or 4 in the half-height kernel (64 rows).
// SB: one 4 KiB piece buffer instead of two (the three-stage half-height kernel has 4 KiB of slice per wave; the LDS
// executes a wave's accesses in order, so re-using the buffer is safe, only the overlap of write and read is lost).
// LNF (EPI_RESID_F32 only): the LayerNorm-fold producer (kernels.h, GemmArgs::ln_part) — the new residual rows also
// leave as operand-dtype rows d = out_new - ln_mean[m] (the next GEMM's A operand) and as per-row partial sums
// (sum d, sum d^2) over the wave's 128 columns; a lane visits row 32 i + 8 it + lane / 8 of the block in the four
// pieces jb = 0..3 of row block i, adds its 16 values in that order and the eight lanes of the row combine by a fixed
// xor butterfly: the statistics are deterministic.
template <typename T, int EPI, bool FULL, int D = 4, bool NT = true, int NMI = 8, bool SB = false, bool LNF = false>
ESMK_DEV void epilogue9_f32(const GemmArgs& p, f32x4 (&acc)[8][NMI], int m_base, int n_base, int lane, char* wl) {
    static_assert(!LNF || EPI == EPI_RESID_F32, "the LayerNorm-fold producer is the residual epilogue");
    constexpr bool lnp = LNF;  // (a runtime form inside the plain residual kernel was measured: the extra code changes the
                               // register allocation of the WHOLE kernel and its main loop — same instruction stream — ran
                               // 3110 - 3170 instead of 2630 - 2790 cycles per K tile, for every residual GEMM:
                               // profiles/r4_ln_fold_ablation.log; so the producer stays an instantiation of its own)
    constexpr int NP = 2 * NMI;  // pieces of 32 x 32
    constexpr int NIB = NMI / 2;  // 32-row blocks
    float s1[4], s2[4];
    float cmv[2][4];  // previous mean of the lane's rows 32 i + 8 it + lane / 8, fetched one 32-row block ahead
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x2 hold[4];  // operand-dtype columns of the pair's first piece
    T* h16 = reinterpret_cast<T*>(p.h16);
    if constexpr (!FULL)
        if (n_base >= p.N || m_base >= p.M) return;  // wave uniform
    float* out = reinterpret_cast<float*>(p.out);
    const int ldc = p.N;
    const int g4 = lane >> 4, l16 = lane & 15;
    f32x4 old[D][4];
    auto load_old = [&](f32x4 (&dst)[4], int piece) ESMK_INL {
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
    auto load_means = [&](int i) ESMK_INL {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = m_base + 32 * i + it * 8 + (lane >> 3);
            cmv[i & 1][it] = (kExperiments && (p.lnf_dbg & 4)) ? 0.f : p.ln_mean[FULL ? m : min(m, p.M - 1)];
        }
    };
    if constexpr (LNF)
        if (lnp) load_means(0);
#pragma unroll
    for (int piece = 0; piece < NP; ++piece) {
        const int i = piece >> 2, jb = piece & 3;
        char* sl = wl + (SB ? 0 : (piece & 1) * 4096);
        if constexpr (LNF) {
            if (jb == 0 && lnp) {
#pragma unroll
                for (int it = 0; it < 4; ++it) s1[it] = s2[it] = 0.f;
                if (i + 1 < NIB) load_means(i + 1);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mi2 = q >> 1, nj2 = q & 1;
            const int row = 16 * mi2 + l16;
            f32x4 v = acc[2 * jb + nj2][2 * i + mi2];
            if constexpr (EPI == EPI_GELU_F32) gelu_fast_x4(v);
            *reinterpret_cast<f32x4*>(sl + row * 128 + (((4 * nj2 + g4) ^ (row & 7)) << 4)) = v;
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
            const bool inside = FULL || (m < p.M && n < p.N);
            if (inside) {
                // non-temporal: a tile's output is not read again by this launch; kept out of the way of the operand
                // panels in the XCD's L2 (profiles/r3_gemm9_mi16_variants.log: +4 .. 7 % on the K = 1280 shapes)
                if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (size_t)m * ldc + n));
                else *reinterpret_cast<f32x4*>(out + (size_t)m * ldc + n) = v;
            }
            if (LNF && lnp) {
                const float c = cmv[i & 1][it];
                float d0 = v[0] - c, d1 = v[1] - c, d2 = v[2] - c, d3 = v[3] - c;
                if (!inside) d0 = d1 = d2 = d3 = 0.f;
                // spelled-out operation order (no compiler contraction choices): every instantiation of this
                // epilogue — full / clipped blocks, both tile heights — produces the same statistics bits
                s1[it] += (d0 + d1) + (d2 + d3);
                s2[it] = __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, __builtin_fmaf(d0, d0, s2[it]))));
                typename Op<T>::v4 pk;
                pk[0] = Op<T>::from(d0), pk[1] = Op<T>::from(d1), pk[2] = Op<T>::from(d2), pk[3] = Op<T>::from(d3);
                // The operand-dtype rows leave as whole 128-byte lines, 16 bytes per lane: a lane holds 4 columns (8 bytes)
                // of piece jb; pieces are taken in pairs (columns 64 (jb / 2) .. + 63 of the row) and neighbouring lanes
                // swap halves — the even lane of a pair stores 8 columns of the first piece, the odd lane 8 columns of the
                // second.  (As 8-byte stores per piece — half lines, two instructions per line — the epilogue took twice
                // its time: profiles/r4_ln_fold_first.log, r4_ln_fold_second.log.)
                const u32x2 cur = __builtin_bit_cast(u32x2, pk);
                if ((jb & 1) == 0) {
                    hold[it] = cur;
                } else {
                    const bool odd = (lane & 1) != 0;
                    const u32x2 prev = hold[it];
                    const u32x2 send = odd ? prev : cur;
                    u32x2 recv;  // quad_perm [1, 0, 3, 2]: the neighbouring lane's words
                    recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xf, 0xf, true);
                    recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xf, 0xf, true);
                    const u32x4 o = odd ? u32x4{recv.x, recv.y, cur.x, cur.y} : u32x4{prev.x, prev.y, recv.x, recv.y};
                    const int col = n_base + (odd ? 32 * jb + 4 * (cc - 1) : 32 * (jb - 1) + 4 * cc);
                    if ((FULL || (m < p.M && col < p.N)) && !(kExperiments && (p.lnf_dbg & 1))) {
                        auto* hd = reinterpret_cast<u32x4*>(h16 + (size_t)m * p.ldh + col);
                        if constexpr (NT) __builtin_nontemporal_store(o, hd);
                        else *hd = o;
                    }
                }
            }
        }
        if constexpr (EPI == EPI_RESID_F32)
            if (piece + D < NP) load_old(old[piece % D], piece + D);
        if constexpr (LNF) {
            if (jb == 3 && lnp && !(kExperiments && (p.lnf_dbg & 2))) {  // the wave's 128 columns of rows 32 i .. 32 i + 31 are complete
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    // the row's eight lanes: lane l adds lanes l-1, then l-2, l-3, then l-4 .. l-7 (DPP row_shr, zeros
                    // shifted in at the start of a 16-lane row) -> lanes 7 and 15 of the row hold the two 8-lane sums
                    float a = s1[it], b = s2[it];
                    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x111, 0xf, 0xf, true));
                    b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x111, 0xf, 0xf, true));
                    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x112, 0xf, 0xf, true));
                    b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x112, 0xf, 0xf, true));
                    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x114, 0xf, 0xf, true));
                    b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x114, 0xf, 0xf, true));
                    const int m = m_base + 32 * i + it * 8 + (lane >> 3);
                    if ((lane & 7) == 7 && (FULL || m < p.M)) {
                        f32x2 st = {a, b};
                        *reinterpret_cast<f32x2*>(p.ln_part + ((size_t)m * p.ln_parts + (n_base >> 7)) * 2) = st;
                    }
                }
            }
        }
    }
}

// operand-dtype outputs (EPI_STORE_T, EPI_GELU_T, EPI_QKV_ROPE): 8 rounds of 32 rows x 64 columns.  A round's 8 KiB
// fp32 image has 256-byte rows, 16-byte chunk c of row r at slot c ^ (r & 7): conflict free for the quad writes
// (8 lanes = 8 rows of one chunk; lane l of a 16 x 16 block holds row l & 15, chunk 4 nq + (l >> 4)) and for the
// row-major reads (16 lanes = 16 different chunks).
// Software pipeline over the rounds: the LDS executes a wave's accesses in order, so round r+1's writes are issued
// right behind round r's reads and land while round r's values go through GELU / RoPE / the conversion — a single
// wave has no partner to hide the LDS round trip behind (15.7k -> cycles of the first version were half latency).
// RR = rows per round: 32 (8 KiB image), or 16 (4 KiB: the three-stage half-height kernel).
// LNF: LayerNorm-fold consumer (kernels.h, GemmArgs::ln_rstd): the accumulators hold raw_row . W''^T (no bias); the value
// is ln_rstd[m] * acc + (bias[n] + bias2[n]) — ONE fma per element; in the q / k epilogue it replaces the multiply by the
// q scale (rstd * scale and (bias + bias2) * scale are formed once per row / per 64-column half).
// X3O (EPI_GELU_T, precision mode f16x3): the rows leave in the hi | hi | lo layout of that mode's operands (GemmArgs::x3_out).
template <typename T, int EPI, bool FULL, bool NT = false, int NMI = 8, int RR = 32, bool LNF = false, bool X3O = false>
ESMK_DEV void epilogue9_t(const GemmArgs& p, f32x4 (&acc)[8][NMI], int m_base, int n_base, int lane, char* wl) {
    static_assert(!X3O || (EPI == EPI_GELU_T && !LNF), "hi | hi | lo output rows: the plain GELU epilogue");
    using V8 = typename Op<T>::v8;
    float bs1[8], bs2[8];  // LNF: bias + bias2 of the lane's columns in the current 64-column half (q: times the scale)
    // LNF: rstd of every row this lane visits (row RR i + RPI it + lane / LPR of the block), all loads in flight at once —
    // one dependent load per row visit cost ~7 k cycles per tile (profiles/r4_ln_fold_first.log)
    constexpr int LPR = EPI == EPI_QKV_ROPE ? 4 : 8;  // lanes per row of a round
    constexpr int RPI = 64 / LPR;                      // rows per 64-lane pass
    constexpr int NIT = RR / RPI;
    constexpr int NI_ = NMI * 16 / RR;
    float rsv[LNF ? NI_ : 1][NIT];
    if constexpr (LNF) {
        if (FULL || m_base < p.M) {
#pragma unroll
            for (int i = 0; i < NI_; ++i)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m = m_base + RR * i + RPI * it + lane / LPR;
                    rsv[i][it] = p.ln_rstd[FULL ? m : min(m, p.M - 1)];
                }
        }
    }
    constexpr int NI = NMI * 16 / RR;  // row blocks of RR rows
    constexpr int NR = 2 * NI;         // rounds: [64-column half][row block]
    constexpr int MB = RR / 16;        // 16-row MFMA blocks per round
    const int g4 = lane >> 4, l16 = lane & 15;
    if constexpr (!FULL)
        if (m_base >= p.M) return;  // wave uniform
    auto write_round = [&](int r) ESMK_INL {
        const int hf = r / NI, i = r % NI;
#pragma unroll
        for (int mi2 = 0; mi2 < MB; ++mi2)
#pragma unroll
            for (int nq = 0; nq < 4; ++nq) {
                const int row = 16 * mi2 + l16, chunk = 4 * nq + g4;
                *reinterpret_cast<f32x4*>(wl + row * 256 + ((chunk ^ (row & 7)) << 4)) = acc[4 * hf + nq][MB * i + mi2];
            }
    };
    // raw[k]: STORE / GELU: slot k / 2 (row (64 (k/2) + lane) / 8, columns 8 (lane & 7) ..), chunk k & 1;
    //         RoPE: slot k / 4 (row (64 (k/4) + lane) / 4, dims 8 (lane & 3) ..), chunks {first half e2 0, 1; second half e2 0, 1}
    auto read_round = [&](f32x4 (&raw)[8]) ESMK_INL {
        if constexpr (EPI == EPI_QKV_ROPE) {
#pragma unroll
            for (int it = 0; it < RR / 16; ++it) {
                const int slot = it * 64 + lane;
                const int r = slot >> 2, g4 = slot & 3;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    raw[4 * it + e2] = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((2 * g4 + e2) ^ (r & 7)) << 4));
                    raw[4 * it + 2 + e2] = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((8 + 2 * g4 + e2) ^ (r & 7)) << 4));
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < RR / 8; ++it) {
                const int slot = it * 64 + lane;
                const int r = slot >> 3, c8 = slot & 7;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2)
                    raw[2 * it + e2] = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((2 * c8 + e2) ^ (r & 7)) << 4));
            }
        }
    };
    auto finish_round = [&](int rd, const f32x4 (&raw)[8]) ESMK_INL {
        const int hf = rd / NI, i = rd % NI;
        const int nb = n_base + 64 * hf;
        if constexpr (!FULL)
            if (nb >= p.N) return;  // wave uniform
        if constexpr (EPI == EPI_QKV_ROPE) {
            // the 64 columns are one head (head_dim 64): dims d and d + 32 rotate together
            // (multihead_attention.py:261 q scaling, rotary_embedding.py:11-20 x*cos + rotate_half(x)*sin)
            const int which = nb / p.E;  // 0 q, 1 k (wave uniform)
            const int head = (nb - which * p.E) >> 6;
            T* qk = reinterpret_cast<T*>(which == 0 ? p.q : p.k);
            const float sc = which == 0 ? p.scaling : 1.0f;
            if constexpr (LNF) {
                if (i == 0) {  // first round of this 64-column half
                    const int c0 = nb + 8 * (lane & 3);
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        f32x4 a = *reinterpret_cast<const f32x4*>(p.bias + c0 + 4 * e2), b = *reinterpret_cast<const f32x4*>(p.bias + c0 + 32 + 4 * e2);
                        if (p.bias2 != nullptr) {
                            const f32x4 a2 = *reinterpret_cast<const f32x4*>(p.bias2 + c0 + 4 * e2), b2 = *reinterpret_cast<const f32x4*>(p.bias2 + c0 + 32 + 4 * e2);
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[e] += a2[e], b[e] += b2[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) bs1[4 * e2 + e] = a[e] * sc, bs2[4 * e2 + e] = b[e] * sc;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < RR / 16; ++it) {
                const int slot = it * 64 + lane;
                const int r = slot >> 2, g4 = slot & 3;  // row of the round, dims [8 g4, 8 g4 + 8)
                const int mm = m_base + RR * i + r;
                const int m = FULL ? mm : min(mm, p.M - 1);
                const int b = m / p.T;
                // rotary position: the row's place in its sequence, or (token-packed batches) in its segment
                const int tt = p.row_pos != nullptr ? p.row_pos[m] : m - b * p.T;
                float rs = sc;
                if constexpr (LNF) rs = rsv[i][it] * sc;
                float y1[8], y2[8];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos + (size_t)tt * 32 + 8 * g4 + 4 * e2);
                    const f32x4 s = *reinterpret_cast<const f32x4*>(p.sin + (size_t)tt * 32 + 8 * g4 + 4 * e2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // the contraction hipcc chose for epilogue8's `a1*c - a2*s`, `a2*c + a1*s`, spelled out:
                        // bit-identical q / k whichever kernel ran (packed == padded == alone stays exact)
                        const float x1 = LNF ? __builtin_fmaf(raw[4 * it + e2][e], rs, bs1[4 * e2 + e]) : raw[4 * it + e2][e] * sc;
                        const float x2 = LNF ? __builtin_fmaf(raw[4 * it + 2 + e2][e], rs, bs2[4 * e2 + e]) : raw[4 * it + 2 + e2][e] * sc;
                        y1[4 * e2 + e] = __builtin_fmaf(x1, c[e], -(x2 * s[e]));
                        y2[4 * e2 + e] = __builtin_fmaf(x2, c[e], x1 * s[e]);
                    }
                }
                V8 o1, o2;
#pragma unroll
                for (int e = 0; e < 8; ++e) o1[e] = Op<T>::from(y1[e]), o2[e] = Op<T>::from(y2[e]);
                if (FULL || mm < p.M) {
                    T* dst = qk + ((size_t)(b * p.H + head) * p.T + (m - b * p.T)) * 64 + 8 * g4;
                    if constexpr (NT) {
                        __builtin_nontemporal_store(o1, reinterpret_cast<V8*>(dst));
                        __builtin_nontemporal_store(o2, reinterpret_cast<V8*>(dst + 32));
                    } else {
                        *reinterpret_cast<V8*>(dst) = o1;
                        *reinterpret_cast<V8*>(dst + 32) = o2;
                    }
                }
            }
        } else {
            T* out = reinterpret_cast<T*>(p.out);
            if constexpr (LNF) {
                if (i == 0) {  // first round of this 64-column half
                    const int c0 = FULL ? nb + 8 * (lane & 7) : min(nb + 8 * (lane & 7), p.N - 8);
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        f32x4 a = *reinterpret_cast<const f32x4*>(p.bias + c0 + 4 * e2);
                        if (p.bias2 != nullptr) {
                            const f32x4 a2 = *reinterpret_cast<const f32x4*>(p.bias2 + c0 + 4 * e2);
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[e] += a2[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) bs1[4 * e2 + e] = a[e];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < RR / 8; ++it) {
                const int slot = it * 64 + lane;
                const int r = slot >> 3, c8 = slot & 7;  // row of the round, columns [8 c8, 8 c8 + 8)
                float v[8];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * e2 + e] = raw[2 * it + e2][e];
                if constexpr (LNF) {
                    const float rs = rsv[i][it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], rs, bs1[e]);
                }
                if constexpr (EPI == EPI_GELU_T) gelu_fast_x8<!X3O>(v);  // four chains: one wave per SIMD has no partner to fill VALU gaps
                                                                        // (X3O keeps hi + lo of the value: the fp32-grade degree-11 set)
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = Op<T>::from(v[e]);
                const int m = m_base + RR * i + r, n = nb + 8 * c8;
                if constexpr (X3O) {
                    if (FULL || (m < p.M && n < p.N)) {
                        V8 lo;
#pragma unroll
                        for (int e = 0; e < 8; ++e) lo[e] = Op<T>::from(v[e] - Op<T>::to(o[e]));
                        T* q = out + (size_t)m * (3 * (size_t)p.N) + (size_t)(n >> 6) * 192 + (n & 63);
                        *reinterpret_cast<V8*>(q) = o;
                        *reinterpret_cast<V8*>(q + 64) = o;
                        *reinterpret_cast<V8*>(q + 128) = lo;
                    }
                } else
                if (FULL || (m < p.M && n < p.N)) {
                    if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<V8*>(out + (size_t)m * p.N + n));
                    else *reinterpret_cast<V8*>(out + (size_t)m * p.N + n) = o;
                }
            }
        }
    };
    write_round(0);
#pragma unroll
    for (int rd = 0; rd < NR; ++rd) {
        f32x4 raw[8];
        read_round(raw);
        if (rd + 1 < NR) write_round(rd + 1);
        __builtin_amdgcn_sched_barrier(0);  // the next round's LDS traffic is issued before this round's arithmetic
        finish_round(rd, raw);
    }
}

// --------------------------------------------------------------------------------------------
// kernel.  VAR = timing experiments (results are wrong): 16 no MFMA, 32 no LDS-DMA, 64 no fragment reads,
// 128 no epilogue.
// --------------------------------------------------------------------------------------------
// HM: half-height tiles (128 x 256; wave blocks 64 x 128) for launches that leave CUs idle with 256-row tiles (small
// batches).  Same LDS image (the activation half of a buffer is half used), 4 + 8 pieces per wave and K tile, 32 MFMA
// slots; every output element sees the same MFMA sequence over K as in the full-height kernel: same bits.
// LNF: LayerNorm fold (kernels.h): EPI_RESID_F32 = producer (second operand-dtype output + row statistics),
// EPI_QKV_ROPE / EPI_V_T / EPI_GELU_T = consumer (accumulators start from 0, row scale + bias in the epilogue).
// X3O: EPI_GELU_T with hi | hi | lo output rows (precision mode f16x3; an instantiation of its own, full height)
template <typename T, int EPI, int VAR = 0, bool HM = false, bool LNF = false, bool X3O = false>
__global__ __launch_bounds__(256, 1) void gemm9_kernel(GemmArgs p, unsigned long long* timing) {
    static_assert(!X3O || (EPI == EPI_GELU_T && !HM && !LNF && VAR == 0), "hi | hi | lo output rows: plain full-height GELU kernel");
    static_assert(!LNF || EPI == EPI_RESID_F32 || EPI == EPI_QKV_ROPE || EPI == EPI_V_T || EPI == EPI_GELU_T || EPI == EPI_QKV_ALL,
                  "LayerNorm fold: epilogue");
    constexpr bool LNC = LNF && EPI != EPI_RESID_F32;  // consumer
    // MFMA orientation of a tile: V^T tiles have a lane own 4 consecutive tokens of one channel.  EPI_QKV_ALL decides per tile
    // (wave uniform) and instantiates the K loop once per orientation.
    using VtDefault = std::integral_constant<bool, EPI == EPI_V_T>;
    constexpr int TM = HM ? 128 : 256;     // tile height
    constexpr int NMI = HM ? 4 : 8;        // 16-row blocks of a wave's block
    constexpr int NPC = NMI + 8;           // DMA pieces per wave and K tile
    constexpr int NRD = NMI + 8;           // fragment reads per K half
    constexpr int NS = 8 * NMI;            // MFMA slots (two MFMAs each) per K tile
    // LDS: full height: two K-tile buffers of 64 KiB + 4 x 8 KiB epilogue slices;  HM: THREE buffers of 48 KiB (128
    // activation rows + 256 weight rows) + 4 x 4 KiB slices — with half the MFMAs per K tile a piece issued two K tiles
    // ahead has less than the MALL round trip to land (small batches are latency bound), three ahead it has.
    constexpr int NST = HM ? 3 : 2;
    constexpr int BUF = HM ? 49152 : Q_BUF, WOFF = HM ? 16384 : Q_WOFF;
    constexpr int SLICE = HM ? 4096 : Q_SLICE;
    static_assert(NST * BUF + 4 * SLICE <= Q_LDS, "LDS budget");
    static_assert(!HM || (VAR & ~(8 | 16 | 32 | 64 | 128)) == 0, "half-height tiles: only the timing-experiment bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Op<T>::v8;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr bool NO_MFMA = (VAR & 16) != 0, NO_DMA = (VAR & 32) != 0, NO_RD = (VAR & 64) != 0, NO_EPI = (VAR & 128) != 0;
    // VAR & 4: odd waves issue their pieces one slot later than even waves (half the TA burst): +-0.
    // VAR & 8: without the s_barrier (results wrong): +3 .. 7 %.
    // VAR & 3 (issue pattern of the 16 pieces): 0 = first barrier at slot 16, one piece per 3 slots (16 .. 61);
    // 1 = slots 24, 25, .. 39 (dense);  2 = 24 .. 61 evenly (one per 2.5 slots);  3 = 24, 26, .. 54.
    // A piece that finds the CU's memory pipeline busy stalls the wave's MFMA issue with it (one wave per SIMD: nobody
    // else issues), and 4 waves x 1 piece per 2 slots is the pipeline's peak rate; measured on the four layer shapes
    // (profiles/r3_gemm9_mi16_variants.log): pattern 0 +8 .. 9 % over 3, 2 in between, 1 -5 %.
    // HM (32 slots, three buffers): Y reads two per slot in slots 0 .. 5, first barrier at 8, the 12 pieces of position
    // s+3 at slots 8, 10, .. 30, second barrier at 16 (the pieces of position s+1 were issued two K tiles ago), X reads of
    // position s+1 one per slot in 16 .. 27.
    constexpr int IMODE = VAR & 3;
    constexpr int M_B1 = HM ? 8 : (IMODE == 0 ? 16 : 24);
    constexpr int M_B2 = HM ? 16 : 52;
    constexpr bool STAGGER = (VAR & 4) != 0;
    constexpr bool NO_BAR = (VAR & 8) != 0;
    // first K tile of a full-height tile: its second K half accumulates through the tied inline-asm MFMA (common.h)
#ifndef ESMK_G9_TIE1
#define ESMK_G9_TIE1 1
#endif
    constexpr bool TIE1 = ESMK_G9_TIE1 && !HM && !NO_MFMA;
    // VAR & 256 / 512 (timing experiments): the K loop of a tile starts at a K offset that depends on the workgroup
    // (256) or on the tile's column block (512) and wraps — the vendor kernel's "StaggerU" against all CUs sweeping the
    // same K offset of their operand rows at once.  256 changes the summation order with the tile -> workgroup map.
    constexpr int KSTAG = (VAR & 256) ? 1 : (VAR & 512) ? 2 : 0;
    // VAR & 1024: only the activation pieces are issued (half the DMA bytes: is the stream bound by bytes or by its
    // round trip?);  VAR & 2048: the pieces carry the non-temporal cache policy.
    constexpr bool HALF_DMA = (VAR & 1024) != 0;
    constexpr int DMA_AUX = (VAR & 2048) ? 2 : 0;
    // pieces of position s+2 issued before the second barrier
    auto piece_slot = [](int k) constexpr {
        if (HM) return 8 + 2 * k;
        return IMODE == 3 ? 24 + 2 * k : IMODE == 1 ? 24 + k : IMODE == 2 ? 24 + (k * 40) / 16 : 16 + 3 * k;
    };
    auto pieces_before = [piece_slot](int m) constexpr {
        int n = 0;
        for (int k = 0; k < NPC; ++k) n += piece_slot(k) < m ? 1 : 0;
        return n;
    };
    constexpr int IN_FLIGHT0 = pieces_before(M_B2);
    constexpr int IN_FLIGHT = HALF_DMA ? (IN_FLIGHT0 + 1) / 2 : IN_FLIGHT0 + (HM ? NPC : 0);  // HM: + position s+2
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nk = p.K >> 6;
    // operand row strides in bytes.  Split weights (f16x2 precision mode, engine.hip): W is the [N, 2 K0] hi | lo image
    // (p.K = 2 K0), the activations keep K0 columns (a_row_bytes) and every activation K tile meets two weight K tiles
    // (a_kt_repeat): the activation K offset advances every other stream position.
    const unsigned rb = (unsigned)p.K * 2u;
    const unsigned rb_a = p.a_row_bytes ? (unsigned)p.a_row_bytes : rb;
    const bool a_rep = p.a_kt_repeat != 0;

    // ---- static persistent schedule (gemm8's: XCD-contiguous ranges of a column-panel blocked tile order) ----
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + 255) >> 8;
    const int total = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int n_my = (cnt > slot) ? (cnt - slot + nslot - 1) / nslot : 0;
    if (n_my == 0) return;
    const int panel_c = p.panel_c > 0 ? p.panel_c : tiles_n;
    const int panel_full = tiles_m * panel_c;
    auto tile_coords = [&](int it, int& tmi, int& tni) ESMK_INL {
        const int o = start + slot + it * nslot;
        const int pnl = o / panel_full;
        const int rem = o - pnl * panel_full;
        const int w = min(panel_c, tiles_n - pnl * panel_c);
        tmi = __builtin_amdgcn_readfirstlane(rem / w);
        tni = __builtin_amdgcn_readfirstlane(pnl * panel_c + (rem - tmi * w));
    };

    // ---- LDS-DMA stream of this wave: 8 + 8 pieces of 8 rows per stream position --------------------------------
    // Scalar state only: one buffer descriptor per operand (base = the tile's panel, num_records = its valid rows), the
    // K offset and the position.  Piece q: per-lane offset (64 wave + lane / 8) rb + swizzled chunk (two registers:
    // the swizzle alternates with q), scalar offset 8 q rb + K offset.
    auto uniform_ptr = [](const void* ptr) ESMK_INL {
        const unsigned long long a = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    __amdgpu_buffer_rsrc_t d_a, d_w;
    int s_kt, s_it;
    unsigned s_koff, s_koff_a;
    const unsigned rb8 = 8u * rb, rb8_a = 8u * rb_a;
    // (row >> 1) & 7 of row = 64 wave + 8 q + lane / 8:  4 (q & 1) + ((lane >> 4) & 3)
    // (HM: the wave stages activation rows [32 wave, + 32) — the same swizzle term — and weight rows [64 wave, + 64))
    const unsigned vo_base = (unsigned)(64 * wave + (lane >> 3)) * rb;
    const unsigned vo_even = vo_base + (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) << 4);
    const unsigned vo_odd = vo_base + (unsigned)(((lane & 7) ^ (4 + ((lane >> 4) & 3))) << 4);
    const unsigned va_base = (unsigned)((HM ? 32 : 64) * wave + (lane >> 3)) * rb_a;
    const unsigned va_even = va_base + (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) << 4);
    const unsigned va_odd = va_base + (unsigned)(((lane & 7) ^ (4 + ((lane >> 4) & 3))) << 4);
    auto set_tile = [&](int it) ESMK_INL {
        int tmi, tni;
        tile_coords(it, tmi, tni);
        s_it = it;
        s_kt = 0;
        if constexpr (KSTAG == 1) s_koff = (unsigned)(((blockIdx.x >> 3) & 3) * (nk >> 2)) * 128u;
        else if constexpr (KSTAG == 2) s_koff = (unsigned)((tni & 3) * (nk >> 2)) * 128u;
        else s_koff = 0;
        s_koff_a = a_rep ? 0u : s_koff;
        const unsigned rows_a = (unsigned)min(TM, p.M - tmi * TM), rows_w = (unsigned)min(256, p.N - tni * 256);
        d_a = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(reinterpret_cast<const char*>(p.A) + (size_t)tmi * TM * rb_a), 0,
                                                (int)__builtin_amdgcn_readfirstlane(rows_a * rb_a), 0x00020000);
        d_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(reinterpret_cast<const char*>(p.W) + (size_t)tni * 256 * rb), 0,
                                                (int)__builtin_amdgcn_readfirstlane(rows_w * rb), 0x00020000);
    };
    auto advance = [&]() ESMK_INL {
        // past the end of the workgroup's tile list the last K tile is re-issued (into a dead buffer), so the
        // wait bookkeeping stays uniform
        if (s_kt + 1 < nk) {
            s_kt = s_kt + 1;
            s_koff += 128u;
            if constexpr (KSTAG != 0)
                if (s_koff == (unsigned)nk * 128u) s_koff = 0;
            s_koff_a = a_rep ? (unsigned)(s_kt >> 1) * 128u : s_koff;
        } else if (s_it + 1 < n_my) {
            set_tile(s_it + 1);
        }
    };
    // piece k = 0..15: operand k & 1 (0 activations, 1 weights), rows 8 (k >> 1) .. of the wave's 64
    // (HM: k = 0..3 activations, rows 8 k .. of the wave's 32; k = 4..11 weights)
    auto issue1 = [&](int k, int buf) ESMK_INL {
        if constexpr (!NO_DMA) {
            if constexpr (HALF_DMA)
                if (k & 1) return;
            const bool isw = HM ? (k >= 4) : ((k & 1) != 0);
            const int q = HM ? (isw ? k - 4 : k) : (k >> 1);
            char* dst = smem + buf * BUF + (isw ? WOFF + wave * 8192 : wave * (HM ? 4096 : 8192)) + q * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isw ? d_w : d_a, (lds_ptr)dst, 16,
                                                     isw ? ((q & 1) ? vo_odd : vo_even) : ((q & 1) ? va_odd : va_even),
                                                     isw ? s_koff + (unsigned)q * rb8 : s_koff_a + (unsigned)q * rb8_a, 0, DMA_AUX);
        }
    };

    // ---- fragments: X = K 0..31, Y = K 32..63 of a K tile (one 16 x 16 x 32 MFMA step each); [16-row block] ---------
    // lane l reads row l & 15 of the block, 16-byte chunk 4 half + (l >> 4) of its 128-byte row
    const int lrow = (lane & 15) * 128;
    const int swz = (lane >> 1) & 7;  // ((row >> 1) & 7 of the block row; blocks start at multiples of 16)
    int xo[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) xo[hf] = ((4 * hf + (lane >> 4)) ^ swz) << 4;
    const int a_off = wr * (NMI * 2048) + lrow;
    const int w_off = WOFF + wc * 16384 + lrow;
    V8 xa[NMI], xw[8], ya[NMI], yw[8];
    if constexpr (NO_RD) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xw[i][e] = yw[i][e] = Op<T>::from(0.f);
                if (i < NMI) xa[i][e] = ya[i][e] = Op<T>::from(0.f);
            }
    }
    // read r = 0 .. NRD - 1 of a half: r < 2 NMI: block r >> 1, r even -> activations, odd -> weights; then the
    // remaining weight blocks
    auto rd1 = [&](V8 (&fa)[NMI], V8 (&fw)[8], const char* bp, int half, int r) ESMK_INL {
        if constexpr (!NO_RD) {
            if (r >= 2 * NMI) {
                const int blk = r - NMI;
                fw[blk] = *reinterpret_cast<const V8*>(bp + w_off + blk * 2048 + xo[half]);
            } else {
                const int blk = r >> 1;
                if (r & 1) fw[blk] = *reinterpret_cast<const V8*>(bp + w_off + blk * 2048 + xo[half]);
                else fa[blk] = *reinterpret_cast<const V8*>(bp + a_off + blk * 2048 + xo[half]);
            }
        }
    };

    f32x4 acc[8][NMI];  // [16-column block][16-row block]
    // bv[nj]: bias of 16-column block nj (the C operand of a tile's first MFMAs: acc = bias + A.W^T, the order
    // gemm8 uses).  Column n = n_base + 16 nj + 4 (lane >> 4) + r sits in register r.  EPI_V_T: the bias varies with
    // the lane, its epilogue adds it.
    f32x4 bv[8];
    // Vector loads (each lane fetches the 4 consecutive columns a register quad holds): issued BEFORE the previous
    // tile's epilogue, so they land under it — the scalar-load form cost ~2k cycles of serialized waits per tile seam.
    auto load_bias = [&](int n_base) ESMK_INL {
        bool done = false;
        if constexpr (EPI != EPI_V_T && !LNC) {
            bool want = p.bias != nullptr;
            if constexpr (EPI == EPI_QKV_ALL) want = want && n_base < 2 * p.E;  // v tiles: the epilogue adds the bias
            if (want) {
                const int g4 = lane >> 4;
                if (n_base + 128 <= p.N) {
#pragma unroll
                    for (int nj = 0; nj < 8; ++nj) bv[nj] = *reinterpret_cast<const f32x4*>(p.bias + n_base + 16 * nj + 4 * g4);
                } else {  // N tail: clamped loads
#pragma unroll
                    for (int nj = 0; nj < 8; ++nj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = n_base + 16 * nj + 4 * g4 + r;
                            bv[nj][r] = n < p.N ? p.bias[n] : 0.f;
                        }
                }
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) bv[nj] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // the two MFMAs of slot m (16 x 16 x 32: 16 cycles each): K half m / (NS / 2), block pair 2 m', 2 m' + 1 of the
    // half's 8 NMI (column block nj = idx / NMI, row block mi = idx % NMI)
    auto mma1 = [&](const V8 (&fa)[NMI], const V8 (&fw)[8], int m, auto first_t, auto vt, auto half_t) ESMK_INL {
        constexpr bool first = decltype(first_t)::value;
        constexpr bool second_half = decltype(half_t)::value;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int idx = 2 * (m % (NS / 2)) + pp;
            const int nj = idx / NMI, mi = idx % NMI;
            f32x4& c = acc[nj][mi];
            const bool use_b = first && m < NS / 2;  // the tile's first K half: C operand = bias
            // the first K tile's second half, full-height tiles: accumulate in place (Op<T>::mma16_tied)
            constexpr bool tied = TIE1 && first && second_half;
            if constexpr (NO_MFMA) {
                asm volatile("" ::"v"(fa[mi]), "v"(fw[nj]));
                if (use_b) c = bv[nj];
            } else if constexpr (TIE1 && first) {
                if constexpr (tied) {
                    if constexpr (decltype(vt)::value) Op<T>::mma16_tied(fa[mi], fw[nj], c);
                    else Op<T>::mma16_tied(fw[nj], fa[mi], c);
                } else if constexpr (decltype(vt)::value) {
                    c = Op<T>::mma16(fa[mi], fw[nj], bv[nj]);
                } else {
                    c = Op<T>::mma16(fw[nj], fa[mi], bv[nj]);
                }
            } else if constexpr (decltype(vt)::value) {  // lane owns 4 consecutive tokens of one channel
                c = use_b ? Op<T>::mma16(fa[mi], fw[nj], bv[nj]) : Op<T>::mma16(fa[mi], fw[nj], c);
            } else {  // lane owns 4 consecutive channels of one token
                c = use_b ? Op<T>::mma16(fw[nj], fa[mi], bv[nj]) : Op<T>::mma16(fw[nj], fa[mi], c);
            }
        }
    };

    int cur = 0;
    auto ktile = [&](auto first, auto vt) ESMK_INL {
        const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
        const char* sb = smem + cur * BUF;
        const char* sn = smem + nxt * BUF;
        advance();  // the stream now stands at position s + NST
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            if (m == M_B1) {  // every wave has read buffer cur completely
                if constexpr (NO_BAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (m == M_B2) {  // every wave's pieces of position s+1 have landed; IN_FLIGHT younger ones stay in flight
                if constexpr (NO_BAR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_FLIGHT) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(IN_FLIGHT) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (m < NS / 2) mma1(xa, xw, m, first, vt, std::false_type{});
            else mma1(ya, yw, m, first, vt, std::true_type{});
            if constexpr (HM) {
                if (m < NRD / 2) {
                    rd1(ya, yw, sb, 1, 2 * m);
                    rd1(ya, yw, sb, 1, 2 * m + 1);
                }
            } else {
                if (m < NRD) rd1(ya, yw, sb, 1, m);
            }
            if constexpr (STAGGER) {  // (with pattern 3) even waves: slots M_B1, M_B1 + 2, ...; odd waves: one slot later
                if (m >= M_B1 && m < M_B1 + 32 && ((m - M_B1) & 1) == 0) {
                    if (!(wave & 1)) issue1((m - M_B1) >> 1, cur);
                } else if (m > M_B1 && m < M_B1 + 33 && ((m - M_B1) & 1) == 1) {
                    if (wave & 1) issue1((m - M_B1 - 1) >> 1, cur);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NPC; ++k)
                    if (piece_slot(k) == m) issue1(k, cur);
            }
            if constexpr (M_B2 + NRD <= NS) {
                if (m >= M_B2 && m < M_B2 + NRD) rd1(xa, xw, sn, 0, m - M_B2);
            } else {  // late second barrier: two reads per slot
                if (m >= M_B2 && m < M_B2 + NRD / 2) {
                    rd1(xa, xw, sn, 0, 2 * (m - M_B2));
                    rd1(xa, xw, sn, 0, 2 * (m - M_B2) + 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // LayerNorm-fold producer, full-height tiles: pin every accumulator quad to the AGPR file across the loop edge.  Left
        // alone, the allocator of this instantiation (512 registers in use) carried ONE quad through the edge in VGPRs:
        // 4 v_accvgpr_write at the top of every K tile and, after the quad's last MFMA, s_nop + 4 v_accvgpr_read — a wait for
        // the matrix pipe in the middle of the issue stream, every K tile.  That, not the epilogue, was the "main loop that
        // runs 20 % slower with the same instructions" of DESIGN.md 4.8: with the pin the loop is the plain kernel's 268
        // instructions (tools/isa_report.py) and fc2 / out-proj as producers cost + 1.7 / + 1.6 ms per step instead of
        // + 6.1 / + 2.9 (B = 64) — the fold now wins at every batch size (profiles/r4_ln_fold_acc_pin.log).
        if constexpr (LNF && EPI == EPI_RESID_F32 && !HM) {
#pragma unroll
            for (int nj = 0; nj < 8; ++nj)
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi) asm volatile("" : "+a"(acc[nj][mi]));
        }
        // K = 64 (one K tile): the epilogue's reads of the accumulators follow the tied MFMAs, which the compiler's hazard
        // recogniser does not see — 16 wait states cover a 4-pass MFMA
        if constexpr (TIE1 && decltype(first)::value) asm volatile("s_nop 7\n\ts_nop 7");
        cur = nxt;
    };

    // ---- prologue: positions 0 and 1 on their way, X fragments of position 0 in registers -------------------------
    set_tile(0);
#pragma unroll
    for (int k = 0; k < NPC; ++k) issue1(k, 0);
#pragma unroll
    for (int st = 1; st < NST; ++st) {
        advance();
#pragma unroll
        for (int k = 0; k < NPC; ++k) issue1(k, st);
    }
    // Residual epilogues are an HBM-bound burst (a 256 KiB read-modify-write per workgroup, every workgroup of the launch
    // at the same moment: 128 MB at the HBM's own rate with the matrix pipes idle — 40 % of the out-projection's tile
    // time, profiles/r3_gemm9_epilogues_final.log).  Tiles all cost the same, so the workgroups stay in lockstep for the
    // whole launch; one group starting late by a fraction of a tile's main loop keeps its bursts under the other group's
    // main loops for every following round.  Pure timing: which tile a workgroup computes, and how, is unchanged.
    if constexpr (EPI == EPI_RESID_F32 && !HM && VAR == 0) {
        if (p.desync > 0) {
            const int g = p.desync_group == 0 ? (int)(blockIdx.x & 1) : p.desync_group == 1 ? (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 3);
            const long long wait = p.desync_group == 2 ? (long long)g * p.desync / 2 : (long long)g * p.desync;
            if (wait > 0) {
                const unsigned long long t0 = __builtin_readcyclecounter();
                while ((long long)(__builtin_readcyclecounter() - t0) < wait) __builtin_amdgcn_s_sleep(32);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(HALF_DMA ? 8 : (NST - 1) * NPC) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NRD; ++r) rd1(xa, xw, smem, 0, r);

    auto stamp = [&](int it, int k) ESMK_INL {
        if (timing != nullptr && tid == 0) {
            unsigned long long* t = timing + ((size_t)blockIdx.x * 32 + (it & 31)) * 4;
            t[k] = __builtin_readcyclecounter();
            if (k == 2) t[3] = wall_clock64();
        }
    };

    {
        int tmi, tni;
        tile_coords(0, tmi, tni);
        load_bias(tni * 256 + wc * 128);
    }
    for (int it = 0; it < n_my; ++it) {
        int tmi, tni;
        tile_coords(it, tmi, tni);
        constexpr int WRM = TM / 2;  // rows of a wave's block
        const int m_base = tmi * TM + wr * WRM, n_base = tni * 256 + wc * 128;
        stamp(it, 0);
        // EPI_QKV_ALL: a tile of columns [2E,3E) is a v tile (tni is wave uniform, 2E a multiple of the tile width)
        bool v_tile = false;
        if constexpr (EPI == EPI_QKV_ALL) v_tile = tni * 256 >= 2 * p.E;
        if constexpr (EPI == EPI_QKV_ALL) {
            if (v_tile) {
                ktile(std::true_type{}, std::true_type{});
#pragma unroll 1
                for (int kt = 1; kt < nk; ++kt) ktile(std::false_type{}, std::true_type{});
            } else {
                ktile(std::true_type{}, std::false_type{});
#pragma unroll 1
                for (int kt = 1; kt < nk; ++kt) ktile(std::false_type{}, std::false_type{});
            }
        } else {
            ktile(std::true_type{}, VtDefault{});
            // (a `.p2align 6` in front of this loop changed nothing, for the fast and the slow instantiations alike:
            // profiles/r4_ln_fold_ablation.log)
#pragma unroll 1
            for (int kt = 1; kt < nk; ++kt) ktile(std::false_type{}, VtDefault{});
        }
        stamp(it, 1);
        // the next tile's bias (bv is dead by now): on its way while this tile's epilogue runs where the epilogue
        // leaves 64 registers free, right behind the epilogue otherwise
        constexpr bool BIAS_EARLY = (EPI == EPI_STORE_T || EPI == EPI_GELU_T);
        auto next_bias = [&]() ESMK_INL {
            if (it + 1 < n_my) {
                int tm2, tn2;
                tile_coords(it + 1, tm2, tn2);
                load_bias(tn2 * 256 + wc * 128);
            }
        };
        if constexpr (BIAS_EARLY) next_bias();
        char* slice = smem + NST * BUF + wave * SLICE;
        const bool full = (m_base + WRM <= p.M) && (n_base + 128 <= p.N);
        // small launches (HM): the next kernel finds the output in the L2 / MALL -> plain stores
        constexpr bool NTS = !HM && (VAR & 4096) == 0;
        if constexpr (NO_EPI) {
#pragma unroll
            for (int nj = 0; nj < 8; ++nj)
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi) asm volatile("" ::"a"(acc[nj][mi]));
        } else if constexpr (EPI == EPI_STORE_F32 || EPI == EPI_GELU_F32 || EPI == EPI_RESID_F32) {
            // residual pieces in flight: 4, or 3 next to the LayerNorm-fold producer's extra state
            constexpr int RD = LNF ? 3 : 4;
            if (full) epilogue9_f32<T, EPI, true, RD, NTS, NMI, HM, LNF>(p, acc, m_base, n_base, lane, slice);
            else epilogue9_f32<T, EPI, false, RD, NTS, NMI, HM, LNF>(p, acc, m_base, n_base, lane, slice);
        } else if constexpr (EPI == EPI_V_T) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int nb = n_base + 64 * hf;
                const bool f2 = (m_base + WRM <= p.M) && (nb + 64 <= p.N) && (p.T % 32 == 0);
                char* sl2 = slice + (HM ? 0 : hf * 4096);
                if (f2) epilogue8m<T, EPI, true, false, false, NMI / 2, 8, NTS, NMI, LNF>(p, acc, 4 * hf, m_base, nb, lane, sl2, 0, 0, 0);
                else epilogue8m<T, EPI, false, false, false, NMI / 2, 8, NTS, NMI, LNF>(p, acc, 4 * hf, m_base, nb, lane, sl2, 0, 0, 0);
            }
        } else if constexpr (EPI == EPI_QKV_ALL) {
            if (v_tile) {  // the v launch's epilogue: columns and bias counted from 2E
                const int col0 = 2 * p.E;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int nb = n_base - col0 + 64 * hf;
                    const bool f2 = (m_base + WRM <= p.M) && (nb + 64 <= p.N - col0) && (p.T % 32 == 0);
                    char* sl2 = slice + (HM ? 0 : hf * 4096);
                    if (f2) epilogue8m<T, EPI_V_T, true, false, false, NMI / 2, 8, NTS, NMI, LNF, true>(p, acc, 4 * hf, m_base, nb, lane, sl2, 0, 0, 0, col0);
                    else epilogue8m<T, EPI_V_T, false, false, false, NMI / 2, 8, NTS, NMI, LNF, true>(p, acc, 4 * hf, m_base, nb, lane, sl2, 0, 0, 0, col0);
                }
            } else {
                if (full) epilogue9_t<T, EPI_QKV_ROPE, true, NTS, NMI, HM ? 16 : 32, LNF>(p, acc, m_base, n_base, lane, slice);
                else epilogue9_t<T, EPI_QKV_ROPE, false, NTS, NMI, HM ? 16 : 32, LNF>(p, acc, m_base, n_base, lane, slice);
            }
        } else {
            if (full) epilogue9_t<T, EPI, true, NTS, NMI, HM ? 16 : 32, LNF, X3O>(p, acc, m_base, n_base, lane, slice);
            else epilogue9_t<T, EPI, false, NTS, NMI, HM ? 16 : 32, LNF, X3O>(p, acc, m_base, n_base, lane, slice);
        }
        if constexpr (!BIAS_EARLY) next_bias();
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

template <typename T, int EPI, int VAR = 0, bool HM = false, bool LNF = false, bool X3O = false>
static hipError_t launch9(GemmArgs p, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm9_kernel<T, EPI, VAR, HM, LNF, X3O>;
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

// LayerNorm fold requested for this call? (producer: EPI_RESID_F32 + ln_part; consumer: q/k, v, fc1 epilogues + ln_rstd)
bool gemm9_ln_fold(const GemmArgs& p, int epi) {
    if (epi == EPI_RESID_F32) return p.ln_part != nullptr;
    if (epi == EPI_QKV_ROPE || epi == EPI_V_T || epi == EPI_GELU_T || epi == EPI_QKV_ALL) return p.ln_rstd != nullptr;
    return false;
}

bool gemm9_supports(const GemmArgs& p, int epi) {
    if (p.K % 64 != 0 || p.N % 8 != 0 || p.M <= 0) return false;
    // of the generalised addressing only the split-weight form (own activation row stride, repeated activation K tiles)
    if (p.w_row_bytes || p.a_kt_bytes || p.w_kt_bytes || p.batch > 1 || p.n_valid > 0 || p.ldc > 0 || p.row_keep != nullptr ||
        p.vt_rows > 0 || p.rowmap_R > 0 || epi == EPI_MSA_CTX || p.head_dim != 64)
        return false;
    if (gemm9_ln_fold(p, epi) && p.a_kt_repeat) return false;  // the fold has no split-weight form
    if (p.a_row_bytes && (p.a_row_bytes % 16 != 0)) return false;
    if ((epi == EPI_QKV_ROPE || epi == EPI_V_T) && p.N % 64 != 0) return false;
    // whole tiles on either side of column 2E; plain (not split) weights
    if (epi == EPI_QKV_ALL && !(p.E > 0 && p.E % 128 == 0 && p.N == 3 * p.E && !p.a_kt_repeat && !p.a_row_bytes && p.bias != nullptr))
        return false;
    if ((long long)256 * p.K * 2 > 0x7fffffffLL) return false;  // a panel must fit a buffer descriptor
    return (epi >= EPI_STORE_T && epi <= EPI_V_T) || epi == EPI_QKV_ALL;
}

template <typename T>
static hipError_t dispatch9(const GemmArgs& p, int epi, int var, hipStream_t st) {
#define ESMK_G9_ALL(V)                                                       \
    switch (epi) {                                                            \
        case EPI_STORE_T: return launch9<T, EPI_STORE_T, V>(p, st);           \
        case EPI_STORE_F32: return launch9<T, EPI_STORE_F32, V>(p, st);       \
        case EPI_GELU_T: return launch9<T, EPI_GELU_T, V>(p, st);             \
        case EPI_GELU_F32: return launch9<T, EPI_GELU_F32, V>(p, st);         \
        case EPI_RESID_F32: return launch9<T, EPI_RESID_F32, V>(p, st);       \
        case EPI_QKV_ROPE: return launch9<T, EPI_QKV_ROPE, V>(p, st);         \
        case EPI_V_T: return launch9<T, EPI_V_T, V>(p, st);                   \
    }
    if (var == 2) { ESMK_G9_ALL(2) }  // A/B of the issue patterns in the whole forward (ESMK_GEMM_IMPL=9:2 / 9:3)
    if (var == 3) { ESMK_G9_ALL(3) }
#undef ESMK_G9_ALL
#ifdef ESMK_EXPERIMENTS
    if constexpr (std::is_same<T, _Float16>::value) {
        if (p.half_m > 0 && epi == EPI_STORE_T) {  // timing experiments on the half-height kernel (results wrong)
            switch (var) {
                case 8: return launch9<T, EPI_STORE_T, 8, true>(p, st);
                case 16: return launch9<T, EPI_STORE_T, 16, true>(p, st);
                case 32: return launch9<T, EPI_STORE_T, 32, true>(p, st);
                case 64: return launch9<T, EPI_STORE_T, 64, true>(p, st);
                case 128: return launch9<T, EPI_STORE_T, 128, true>(p, st);
                case 96: return launch9<T, EPI_STORE_T, 96, true>(p, st);
                case 224: return launch9<T, EPI_STORE_T, 224, true>(p, st);
            }
        }
    }
#endif
    if (p.x3_out) {  // f16x3: fc1 + GELU with hi | hi | lo output rows — full height, fp16
        if constexpr (std::is_same<T, _Float16>::value) {
            if (epi == EPI_GELU_T && var == 0 && !gemm9_ln_fold(p, epi)) return launch9<T, EPI_GELU_T, 0, false, false, true>(p, st);
        }
        return hipErrorInvalidValue;
    }
    if (gemm9_ln_fold(p, epi)) {  // LayerNorm fold: producer / consumer forms of the four epilogues, both tile heights
        if (var != 0) return hipErrorInvalidValue;
        if (epi == EPI_RESID_F32 && (p.h16 == nullptr || p.ln_mean == nullptr || p.ldh < p.N || p.ln_parts < (p.N + 127) / 128))
            return hipErrorInvalidValue;
        if (epi != EPI_RESID_F32 && p.bias == nullptr) return hipErrorInvalidValue;
        if (p.half_m > 0) {
            switch (epi) {
                case EPI_RESID_F32: return launch9<T, EPI_RESID_F32, 0, true, true>(p, st);
                case EPI_QKV_ROPE: return launch9<T, EPI_QKV_ROPE, 0, true, true>(p, st);
                case EPI_V_T: return launch9<T, EPI_V_T, 0, true, true>(p, st);
                case EPI_GELU_T: return launch9<T, EPI_GELU_T, 0, true, true>(p, st);
                case EPI_QKV_ALL: return launch9<T, EPI_QKV_ALL, 0, true, true>(p, st);
            }
        }
        switch (epi) {
            case EPI_RESID_F32: return launch9<T, EPI_RESID_F32, 0, false, true>(p, st);
            case EPI_QKV_ROPE: return launch9<T, EPI_QKV_ROPE, 0, false, true>(p, st);
            case EPI_V_T: return launch9<T, EPI_V_T, 0, false, true>(p, st);
            case EPI_GELU_T: return launch9<T, EPI_GELU_T, 0, false, true>(p, st);
            case EPI_QKV_ALL: return hipErrorInvalidValue;  // half-height tiles only (see gemm_qkv_one_launch)
        }
        return hipErrorInvalidValue;
    }
    if (var == 0 && p.half_m > 0) {  // half-height tiles
        switch (epi) {
            case EPI_STORE_T: return launch9<T, EPI_STORE_T, 0, true>(p, st);
            case EPI_STORE_F32: return launch9<T, EPI_STORE_F32, 0, true>(p, st);
            case EPI_GELU_T: return launch9<T, EPI_GELU_T, 0, true>(p, st);
            case EPI_GELU_F32: return launch9<T, EPI_GELU_F32, 0, true>(p, st);
            case EPI_RESID_F32: return launch9<T, EPI_RESID_F32, 0, true>(p, st);
            case EPI_QKV_ROPE: return launch9<T, EPI_QKV_ROPE, 0, true>(p, st);
            case EPI_V_T: return launch9<T, EPI_V_T, 0, true>(p, st);
            case EPI_QKV_ALL: return launch9<T, EPI_QKV_ALL, 0, true>(p, st);
        }
    }
    if (var == 0) {
        switch (epi) {
            case EPI_STORE_T: return launch9<T, EPI_STORE_T>(p, st);
            case EPI_STORE_F32: return launch9<T, EPI_STORE_F32>(p, st);
            case EPI_GELU_T: return launch9<T, EPI_GELU_T>(p, st);
            case EPI_GELU_F32: return launch9<T, EPI_GELU_F32>(p, st);
            case EPI_RESID_F32: return launch9<T, EPI_RESID_F32>(p, st);
            case EPI_QKV_ROPE: return launch9<T, EPI_QKV_ROPE>(p, st);
            case EPI_V_T: return launch9<T, EPI_V_T>(p, st);
            case EPI_QKV_ALL: return hipErrorInvalidValue;  // half-height tiles only (see gemm_qkv_one_launch)
        }
    }
    if constexpr (std::is_same<T, _Float16>::value) {
        if (epi == EPI_STORE_T) {  // timing experiments (tools/bench_gemm9.py --dbg)
            switch (var) {  // schedule variants: same results
                case 1: return launch9<T, EPI_STORE_T, 1>(p, st);
                case 7: return launch9<T, EPI_STORE_T, 7>(p, st);
                case 512: return launch9<T, EPI_STORE_T, 512>(p, st);    // K stagger by column block
                case 2048: return launch9<T, EPI_STORE_T, 2048>(p, st);  // non-temporal operand loads
                case 4096: return launch9<T, EPI_STORE_T, 4096>(p, st);  // plain (temporal) stores
#ifdef ESMK_EXPERIMENTS  // parts of the kernel removed: results wrong
                case 8: return launch9<T, EPI_STORE_T, 8>(p, st);
                case 16: return launch9<T, EPI_STORE_T, 16>(p, st);
                case 32: return launch9<T, EPI_STORE_T, 32>(p, st);
                case 64: return launch9<T, EPI_STORE_T, 64>(p, st);
                case 128: return launch9<T, EPI_STORE_T, 128>(p, st);
                case 96: return launch9<T, EPI_STORE_T, 96>(p, st);
                case 224: return launch9<T, EPI_STORE_T, 224>(p, st);
                case 1040: return launch9<T, EPI_STORE_T, 1040>(p, st);  // no MFMAs, half the DMA bytes
#endif
            }
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemm9(const GemmArgs& p, int epi, int operand_dtype, int var, hipStream_t st) {
    if (!gemm9_supports(p, epi)) return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_F16) return dispatch9<_Float16>(p, epi, var, st);
    if (operand_dtype == ESMK_DT_BF16) return dispatch9<__bf16>(p, epi, var, st);
    return hipErrorInvalidValue;
}

}  // namespace esmk
