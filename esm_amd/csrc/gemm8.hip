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
//     instructions + a COUNTED s_waitcnt vmcnt(8)) and a matrix section (16 v_mfma_f32_16x16x32 =
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
#include "gemm_epi.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace esmk {

// --------------------------------------------------------------------------------------------
// kernel
// --------------------------------------------------------------------------------------------
// SCHED 0: fragment reads per load section 12 / 4 / 8 / 0.   SCHED 1: 4 / 4 / 8 / 8 — the (i0,i1)
// activation fragments of the NEXT stream position are read in section L3 into their own registers.
// DBG != 0 builds timing experiments (tools/microbench.py --only dbg8; results are wrong):
//   bit 0 no MFMA, bit 1 no LDS-DMA, bit 2 no fragment reads, bit 3 no epilogue,
//   bit 4 epilogue without its global stores, bit 5 leave a FULL epilogue's stores in flight behind the
//   next K tile (vmcnt(8 + S) for its waits; measured slower than waiting), bit 6 every DMA re-reads
//   K slab 0 (L2 resident).
// PF > 0: every wave touches one dword of 64 of the 512 cache lines of stream position s + PF per K
// tile (an L2 prefetch: the LDS-DMA of that position then hits the XCD's L2 instead of paying the
// fabric / HBM latency inside the two-K-tile-deep DMA window).
// GEN: generalised addressing (strides, batch, row remaps; GemmArgs fields after `scaling`) — a separate
// instantiation so that the dense layer-stack kernels keep their register budget.
// HM: half-height tiles (128 x 256).  Everything — LDS units, DMA stream, waits, barriers — stays as it is; the
// two wave groups own 64 rows each instead of 128, so the matrix sections of phases 2 and 3 (fragments i2, i3) are
// empty and unit U3 re-stages the rows of U0 (L2 hits, never read).  A tile then costs 0.7 - 0.95 of a full one but
// there are twice as many: used when 256-row tiles leave most CUs without work (small batches: B = 4 x 1024 tokens has
// 80 tiles for 256 CUs at N = 1280).  Every output element still sees the same MFMA sequence over K, so the results
// are bit-identical to the full-height kernel — independent of the batch size.
template <typename T, int EPI, int SCHED = 0, int DBG = 0, int PF = 0, bool GEN = false, bool HM = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs p, unsigned long long* timing) {
    static_assert(!HM || (SCHED == 0 && !GEN), "half-height tiles: dense layer-stack kernels, schedule 0 only");
    constexpr int TM = HM ? 128 : 256;  // tile height
    constexpr int GM = TM / 2;          // rows per wave group
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Op<T>::v8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // M half of the tile AND ping-pong group
    const int wn = wave & 3;    // 64-column slice of the tile
    const int nk = p.K >> 6;
    const unsigned a_rb = (GEN && p.a_row_bytes) ? (unsigned)p.a_row_bytes : (unsigned)p.K * 2u;  // operand row strides
    const unsigned w_rb = (GEN && p.w_row_bytes) ? (unsigned)p.w_row_bytes : (unsigned)p.K * 2u;
    const long long a_kt = (GEN && p.a_kt_bytes) ? p.a_kt_bytes : 128, w_kt = (GEN && p.w_kt_bytes) ? p.w_kt_bytes : 128;
    const int n_valid = (GEN && p.n_valid > 0) ? p.n_valid : p.N;
    const int nbatch = GEN ? p.batch : 1, binner = GEN ? p.batch_inner : 1;

    // ---- static persistent schedule -----------------------------------------------------------
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + 255) >> 8;
    const int tiles_mn = tiles_m * tiles_n;
    const int total = tiles_mn * nbatch;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int n_my = (cnt > slot) ? (cnt - slot + nslot - 1) / nslot : 0;
    if (n_my == 0) return;
    const int panel_c = p.panel_c > 0 ? p.panel_c : tiles_n;
    const int panel_full = tiles_m * panel_c;
    auto tile_coords = [&](int it, int& tmi, int& tni, int& bz) {
        const int og = start + slot + it * nslot;
        bz = __builtin_amdgcn_readfirstlane(nbatch > 1 ? og / tiles_mn : 0);
        const int o = og - bz * tiles_mn;
        const int pnl = o / panel_full;
        const int rem = o - pnl * panel_full;
        const int w = min(panel_c, tiles_n - pnl * panel_c);
        // (integer division runs on the VALU: pin the wave-uniform results back into SGPRs)
        tmi = __builtin_amdgcn_readfirstlane(rem / w);
        tni = __builtin_amdgcn_readfirstlane(pnl * panel_c + (rem - tmi * w));
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
        int tmi, tni, bz;
        tile_coords(it, tmi, tni, bz);
        const int zo = bz / binner, zi = bz - zo * binner;
        const long long a_boff = GEN ? zo * p.a_bo + zi * p.a_bi : 0, w_boff = GEN ? zo * p.w_bo + zi * p.w_bi : 0;
        s.it = it;
        s.kt = 0;
        if (unit == 0 || unit == 3) {  // A rows: group (ur>>6) * 128 + (unit 3 ? 64 : 0) + (ur & 63)
            const int lim = p.M - tmi * TM - 1;
            const int add = (unit == 3 && !HM) ? 64 : 0;  // HM: U3 repeats the rows of U0
            s.base = reinterpret_cast<const char*>(p.A) + (size_t)tmi * TM * a_rb + a_boff;
            s.off0 = (unsigned)min((ur0 >> 6) * GM + add + (ur0 & 63), lim) * a_rb + cs0;
            s.off1 = (unsigned)min((ur1 >> 6) * GM + add + (ur1 & 63), lim) * a_rb + cs1;
        } else {  // W rows: wave column (ur>>5) * 64 + (unit 2 ? 32 : 0) + (ur & 31)
            const int lim = n_valid - tni * 256 - 1;
            const int add = unit == 2 ? 32 : 0;
            s.base = reinterpret_cast<const char*>(p.W) + (size_t)tni * 256 * w_rb + w_boff;
            s.off0 = (unsigned)max(min((ur0 >> 5) * 64 + add + (ur0 & 31), lim), 0) * w_rb + cs0;
            s.off1 = (unsigned)max(min((ur1 >> 5) * 64 + add + (ur1 & 31), lim), 0) * w_rb + cs1;
            if (lim < 0) s.base -= (size_t)tni * 256 * w_rb - (size_t)max(n_valid - 1, 0) * w_rb;  // whole tile past n_valid
        }
    };
    auto issue = [&](Stream& s, int unit, int buf) {
        char* dst = smem + buf * P_BUF + unit * P_UNIT + wave * 2048;
        if constexpr (!(DBG & 2)) {
            glds16(s.base + s.off0, dst);
            glds16(s.base + s.off1, dst + 1024);
        }
        // advance to the next stream position; past the end of the workgroup's tile list the last
        // K tile is re-issued (into a dead buffer) so the vmcnt bookkeeping stays uniform
        if (s.kt + 1 < nk) {
            s.kt = __builtin_amdgcn_readfirstlane(s.kt + 1);
            if constexpr (!(DBG & 64)) {
                if (unit == 0 || unit == 3) {
                    // split weights: the activations' K tile kt / 2 meets W_hi (kt even) and W_lo (kt odd)
                    if (!(GEN && p.a_kt_repeat) || !(s.kt & 1)) s.base += a_kt;
                } else {
                    s.base += w_kt;
                }
            }
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

    // ---- L2 prefetch stream (PF > 0): waves 0-3 cover the 256 A rows, waves 4-7 the 256 W rows ----
    Stream sP;
    unsigned pf_dummy = 0;  // destination of the prefetch loads; never read
    auto set_tile_pf = [&](int it) {
        int tmi, tni, bz;
        tile_coords(it, tmi, tni, bz);
        sP.it = it;
        sP.kt = 0;
        const int row = (wave & 3) * 64 + lane;
        if (wave < 4) {
            sP.base = reinterpret_cast<const char*>(p.A) + (size_t)tmi * TM * a_rb;  // (HM: rows 128.. are the next tile's)
            sP.off0 = (unsigned)min(row, p.M - tmi * TM - 1) * a_rb;
        } else {
            sP.base = reinterpret_cast<const char*>(p.W) + (size_t)tni * 256 * w_rb;
            sP.off0 = (unsigned)min(row, p.N - tni * 256 - 1) * w_rb;
        }
    };
    auto advance_pf = [&]() {
        if (sP.kt + 1 < nk) {
            sP.kt = __builtin_amdgcn_readfirstlane(sP.kt + 1);
            sP.base += 128;
        } else if (sP.it + 1 < n_my) {
            set_tile_pf(sP.it + 1);
        }
    };
    auto prefetch = [&]() {
        if constexpr (PF > 0) {
            // hipcc does not count an asm load: pf_dummy stays reserved ("+v" below) until a later counted
            // wait has retired it (vmcnt is in order), and nothing ever consumes the value
            asm volatile("global_load_dword %0, %1, %2" : "=v"(pf_dummy) : "v"(sP.off0), "s"(sP.base) : "memory");
            advance_pf();
        }
    };
    if constexpr (PF > 0) {
        set_tile_pf(0);
#pragma unroll 1
        for (int k = 0; k < PF; ++k) advance_pf();
    }

    // ---- fragment read offsets (16 x 16 x 32 MFMA blocks: lane l reads row l & 15 of a 16-row block, 16-byte chunk
    // 4 ks + (l >> 4) of its 128-byte row; ks = K half of the tile) ----------------------------------------------
    const int lrow = (lane & 15) * 128;
    const int swz = (lane >> 1) & 7;
    int xo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xo[ks] = ((4 * ks + (lane >> 4)) ^ swz) << 4;
    const int a_off = grp * (64 * 128) + lrow;
    const int w_off = wn * (32 * 128) + lrow;

    V8 fa[4][2], fb[4][2], fw0[2][2], fw1[2][2];  // [16-row / 16-column block][K half]; fb: rows 64..127, SCHED 1 only
    f32x4 acc[4][8];                              // [16-column block of the wave's 64][16-row block of its 128]
    if constexpr (DBG & 4) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int b = 0; b < 4; ++b) fa[b][ks][e] = fb[b][ks][e] = Op<T>::from(0.f);
                fw0[0][ks][e] = fw0[1][ks][e] = fw1[0][ks][e] = fw1[1][ks][e] = Op<T>::from(0.f);
            }
    }

    auto rdA = [&](V8 (&f)[4][2], const char* ub) {
        if constexpr (DBG & 4) return;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f[b][ks] = *reinterpret_cast<const V8*>(ub + a_off + b * 2048 + xo[ks]);
    };
    auto rdW = [&](V8 (&fw)[2][2], const char* ub) {
        if constexpr (DBG & 4) return;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fw[b][ks] = *reinterpret_cast<const V8*>(ub + w_off + b * 2048 + xo[ks]);
    };
    // one 64 x 32 quadrant over K = 64: 16 MFMAs on 8 accumulators (column blocks 2 jb, 2 jb + 1; row blocks 4 ib ..)
    auto quad = [&](int jb, int ib, const V8 (&fw)[2][2], const V8 (&f)[4][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    f32x4& c = acc[2 * jb + nb][4 * ib + mb];
                    if constexpr (DBG & 1) {
                        asm volatile("" ::"v"(fw[nb][ks]), "v"(f[mb][ks]));
                    } else if constexpr (EPI == EPI_V_T) {  // lane owns 4 consecutive tokens of one channel
                        c = Op<T>::mma16(f[mb][ks], fw[nb][ks], c);
                    } else {  // lane owns 4 consecutive channels of one token
                        c = Op<T>::mma16(fw[nb][ks], f[mb][ks], c);
                    }
                }
        __builtin_amdgcn_s_setprio(0);
    };

    int cur = 0;  // LDS buffer of the current stream position
    // acc = bias (the nn.Linear bias rides through the K loop; fp32).  The wave's 64 bias values are
    // fetched with SCALAR loads through the constant address space: they do not enter the vmcnt queue,
    // so they neither wait for the LDS-DMA stream nor for the previous epilogue's stores.
    auto init_acc = [&](int n_base) {
        bool done = false;
        if constexpr (EPI != EPI_V_T) {
            if (p.bias != nullptr) {
                const int g4 = lane >> 4;  // the lane's columns of a 16-column block: 4 g4 .. 4 g4 + 3
                if (n_base + 64 <= p.N) {
                    typedef const __attribute__((address_space(4))) f32x16* cvec_ptr;
                    cvec_ptr cb = (cvec_ptr)(unsigned long long)(p.bias + n_base);
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) {
                        // 16 SGPRs at a time (the whole row would pin 64 SGPRs next to the loop state): one s_load_dwordx16
                        const f32x16 bb = cb[nj];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // two-level select on the bits of g4 over OPAQUE scalars (hipcc folds a visible select chain
                            // into a dynamic vector index — 15 selects per value — or into per-lane global loads)
                            float e0 = bb[r], e1 = bb[4 + r], e2 = bb[8 + r], e3 = bb[12 + r];
                            asm volatile("" : "+s"(e0), "+s"(e1), "+s"(e2), "+s"(e3));
                            const float lo = (lane & 16) ? e1 : e0, hi = (lane & 16) ? e3 : e2;
                            const float b = (lane & 32) ? hi : lo;
#pragma unroll
                            for (int mi = 0; mi < 8; ++mi) acc[nj][mi][r] = b;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {  // N tail: clamped vector loads
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = n_base + 16 * nj + 4 * g4 + r;
                            const float b = n < p.N ? p.bias[n] : 0.f;
#pragma unroll
                            for (int mi = 0; mi < 8; ++mi) acc[nj][mi][r] = b;
                        }
                }
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) acc[nj][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // One K tile = 4 phases.  WN = vmcnt immediate of the three counted waits: 8 in steady state
    // (a unit is waited for 4 load sections = 8 DMA instructions after its issue), 8 + S for the first
    // K tile after a FULL epilogue, whose S store instructions are younger than every DMA the tile
    // waits for: they stay in flight behind the main loop instead of stalling it.
    // EPI_RESID_F32: the epilogue reads the fp32 residual tile it adds to.  Its 8 pieces are latency
    // bound on those loads (HBM miss ~2.5k cycles each), so at the end of the tile's LAST K tile every
    // lane touches one dword of 4 of the wave's 256 residual cache lines: by the time the epilogue asks
    // for them they sit in the XCD's L2.  Measured (profiles/r1_v4_*): fc2 (80 K tiles per tile) -3 %,
    // out_proj (20 K tiles) +7 % because its epilogue burst is HBM-bandwidth bound, so it is only used for
    // long K loops.  (DBG bit 7 switches it off for A/B runs.)
    constexpr bool XPF = (EPI == EPI_RESID_F32) && !(DBG & 128);
    unsigned x_dummy = 0;  // destination of those loads; never read
    int m_base_cur = 0, n_base_cur = 0;
    bool last_kt = false;
    auto prefetch_x = [&]() {
        if constexpr (XPF) {
            const float* xo_ = reinterpret_cast<const float*>(p.out);
#pragma unroll
            for (int q = 0; q < (HM ? 2 : 4); ++q) {  // 2 cache lines per row of the wave's 128 (HM: 64) rows
                const int L = q * 64 + lane;
                const int row = min(m_base_cur + (L >> 1), p.M - 1);
                const int n = min(n_base_cur + (L & 1) * 32, p.N - 4);
                const float* addr = xo_ + (size_t)remap_row<GEN>(p, row) * ((GEN && p.ldc > 0) ? p.ldc : p.N) + n;
                asm volatile("global_load_dword %0, %1, off" : "+v"(x_dummy) : "v"(addr) : "memory");
            }
        }
    };
    constexpr int S_EPI = epilogue_stores<EPI>();
    constexpr int WBASE = PF > 0 ? 9 : 8;  // VM instructions a wave issues per K tile: 8 DMA (+ 1 prefetch)
    bool young_stores = false;  // wave uniform: this K tile directly follows a FULL epilogue
    auto wait_units = [&]() {
        if ((DBG & 32) != 0 && young_stores) wait_vmcnt<WBASE + S_EPI>();
        else wait_vmcnt<WBASE>();
    };
    auto ktile = [&]() {
        const char* sb = smem + cur * P_BUF;
        if constexpr (SCHED == 0) {
            // ---- phase 0 -----------------------------------------------------------------------
            rdA(fa, sb);
            rdW(fw0, sb + P_UNIT);
            issue(sW1, 2, cur ^ 1);
            wait_units();
            wg_barrier();
            quad(0, 0, fw0, fa);
            wg_barrier();
            // ---- phase 1 -----------------------------------------------------------------------
            rdW(fw1, sb + 2 * P_UNIT);
            issue(sA1, 3, cur ^ 1);
            wait_units();
            wg_barrier();
            quad(1, 0, fw1, fa);
            wg_barrier();
            // ---- phase 2 -----------------------------------------------------------------------
            if constexpr (!HM) rdA(fa, sb + 3 * P_UNIT);
            issue(sA0, 0, cur);
            wg_barrier();
            if constexpr (!HM) quad(1, 1, fw1, fa);
            wg_barrier();
            // ---- phase 3 -----------------------------------------------------------------------
            issue(sW0, 1, cur);
            wait_units();
            if constexpr (PF > 0) asm volatile("" : "+v"(pf_dummy));  // the previous prefetch has retired
            prefetch();
            if constexpr (XPF)
                if (last_kt) prefetch_x();  // after the K tile's last counted wait (last_kt: the K tile chosen for it)
            wg_barrier();
            if constexpr (!HM) quad(0, 1, fw0, fa);
            wg_barrier();
        } else {
            // fa = (i0,i1) fragments of this position, read in L3 of the previous position
            rdW(fw0, sb + P_UNIT);
            issue(sW1, 2, cur ^ 1);
            wait_units();  // U2 of cur
            wg_barrier();
            quad(0, 0, fw0, fa);
            wg_barrier();
            rdW(fw1, sb + 2 * P_UNIT);
            issue(sA1, 3, cur ^ 1);
            wait_units();  // U3 of cur
            wg_barrier();
            quad(1, 0, fw1, fa);
            wg_barrier();
            rdA(fb, sb + 3 * P_UNIT);
            issue(sA0, 0, cur);
            wait_units();  // U0 of cur^1 (next position)
            wg_barrier();
            quad(1, 1, fw1, fb);
            wg_barrier();
            rdA(fa, smem + (cur ^ 1) * P_BUF);
            issue(sW0, 1, cur);
            wait_units();  // U1 of cur^1
            wg_barrier();
            quad(0, 1, fw0, fb);
            wg_barrier();
        }
        cur ^= 1;
    };

    wait_vmcnt8();  // U0, U1 of position 0 have landed
    wg_barrier();
    if constexpr (SCHED == 1) rdA(fa, smem);
    auto stamp = [&](int it, int k) {
        if (timing != nullptr && tid == 0) {
            unsigned long long* t = timing + ((size_t)blockIdx.x * 32 + (it & 31)) * 4;
            t[k] = __builtin_readcyclecounter();
            if (k == 2) t[3] = wall_clock64();  // constant-rate counter: ticks / wall gives the shader clock
        }
    };

    for (int it = 0; it < n_my; ++it) {
        int tmi, tni, bz;
        tile_coords(it, tmi, tni, bz);
        const int zo = bz / binner, zi = bz - zo * binner;
        const size_t out_off = GEN ? (size_t)(zo * p.o_bo + zi * p.o_bi) : 0;
        const int m_base = tmi * TM + grp * GM, n_base = tni * 256 + wn * 64;
        init_acc(n_base);
        stamp(it, 0);

        if (grp == 1) wg_barrier();  // group 1 runs one section behind group 0
        m_base_cur = m_base;
        n_base_cur = n_base;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            last_kt = (kt == p.xpf_kt);  // K tile after which the residual tile is prefetched (-1: never)
            ktile();
            young_stores = false;
        }
        if (grp == 0) wg_barrier();  // re-align: both groups run the epilogue together
        stamp(it, 1);

        char* slice = smem + P_EPI + wave * P_SLICE;
        if constexpr (DBG & 8) {
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) asm volatile("" ::"v"(acc[nj][mi]));
        } else {
            bool full = (m_base + GM <= p.M) && (n_base + 64 <= p.N);
            if constexpr (EPI == EPI_V_T) full = full && (p.T % 32 == 0);
            constexpr int NI = HM ? 2 : 4;
            if (full) epilogue8m<T, EPI, true, (DBG & 16) != 0, GEN, NI>(p, acc, 0, m_base, n_base, lane, slice, out_off, zo, zi);
            else epilogue8m<T, EPI, false, (DBG & 16) != 0, GEN, NI>(p, acc, 0, m_base, n_base, lane, slice, out_off, zo, zi);
            young_stores = full && !(DBG & 16) && !HM;
        }
        if constexpr (XPF) asm volatile("" : "+v"(x_dummy));  // the epilogue's own loads retired them
        stamp(it, 2);
    }
    wait_vmcnt0();  // the trailing (dummy) DMA writes must land before the LDS is released
    if constexpr (PF > 0) asm volatile("" ::"v"(pf_dummy));
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
// generalised addressing requested?  (MSA Transformer calls, batched / strided / remapped GEMMs, packed batches)
bool gemm8_generalised(const GemmArgs& p, int epi) {
    return p.a_row_bytes || p.w_row_bytes || p.a_kt_bytes || p.w_kt_bytes || p.a_kt_repeat || p.batch > 1 || p.n_valid > 0 ||
           p.ldc > 0 || p.row_keep != nullptr || p.vt_rows > 0 || p.rowmap_R > 0 || epi == EPI_MSA_CTX ||
           p.head_dim != 64 || p.row_pos != nullptr;
}

static int num_workgroups();

// Half-height tiles (gemm8_kernel<..., HM>) cost 0.7 - 0.95 of a full tile each (tools/bench_half_tiles.py,
// profiles/r2_half_height_tiles.log: with half the MFMAs the K step lands on the loop's ~1700-cycle non-MFMA
// skeleton — DMA landing, fragment reads, barriers — instead of 1041 cycles of MFMA issue), so they only pay when BOTH tile heights fit in the same number of rounds over the
// CUs, i.e. when 256-row tiles leave CUs idle (B <= 4 sequences of 1024 tokens at N = 1280).  ESMK_GEMM8_HM = 0 / 1
// forces.
constexpr double HM_TILE_COST = 0.8;
bool gemm8_half_height(const GemmArgs& p) {
    if (p.half_m != 0) return p.half_m > 0;
    static const int env = [] {
        const char* e = getenv("ESMK_GEMM8_HM");
        return e ? atoi(e) : -1;
    }();
    if (env >= 0) return env != 0;
    const long long wg = num_workgroups(), tn = (p.N + 255) / 256;
    const long long t256 = (long long)((p.M + 255) / 256) * tn, t128 = (long long)((p.M + 127) / 128) * tn;
    const double full = (double)((t256 + wg - 1) / wg), half = HM_TILE_COST * (double)((t128 + wg - 1) / wg);
    return half < 0.97 * full;
}

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

static unsigned long long* g_timing = nullptr;  // esmk_debug_gemm_timing()
void gemm8_set_timing(unsigned long long* dev_buf) { g_timing = dev_buf; }

constexpr int PF_DEFAULT = 0;

template <typename T, int EPI, int SCHED = 0, int DBG = 0, int PF = PF_DEFAULT, bool GEN = false, bool HM = false>
static hipError_t launch8(GemmArgs p, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm8_kernel<T, EPI, SCHED, DBG, PF, GEN, HM>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_n = (p.N + 255) / 256;
    static const int env_panel = [] {
        const char* e = getenv("ESMK_PANEL_C");
        return e ? atoi(e) : 0;
    }();
    if (p.panel_c <= 0 && env_panel > 0) p.panel_c = env_panel < tiles_n ? env_panel : tiles_n;
    if (p.panel_c <= 0) {
        // 32 concurrent tiles per XCD should form a block as square as possible: ~6 x 5 or 8 x 4
        if (tiles_n <= 6) p.panel_c = tiles_n;
        else if (tiles_n % 5 == 0) p.panel_c = 5;
        else if (tiles_n % 4 == 0) p.panel_c = 4;
        else if (tiles_n % 6 == 0) p.panel_c = 6;
        else p.panel_c = 5;
    }
    {   // EPI_RESID_F32: K tile after which each wave touches its residual lines (see prefetch_x).  The prefetch pays
        // when a workgroup has ONE tile (small batches: nothing else hides the epilogue's residual loads — fc2 at B = 4:
        // 4.26 -> 3.38 ms per forward) and costs when every workgroup walks several tiles and the chip is bandwidth /
        // power bound (B = 64, round 2, one call: fc2 26.2 -> 25.3 ms per step without it, out_proj 8.75 -> 9.7-10.0 with
        // it), so it is used for single-round launches with a long K loop only.
        // ESMK_XPF_D / ESMK_XPF_MIN_NK / ESMK_XPF_ROUNDS (experiments): distance from the end of the K loop, shortest K
        // loop, most tile rounds it is used for.
        static const int xd = [] { const char* e = getenv("ESMK_XPF_D"); return e ? atoi(e) : 1; }();
        static const int xmin = [] { const char* e = getenv("ESMK_XPF_MIN_NK"); return e ? atoi(e) : 40; }();
        static const int xrounds = [] { const char* e = getenv("ESMK_XPF_ROUNDS"); return e ? atoi(e) : 1; }();
        const int nk = p.K / 64, wg = num_workgroups();
        const long long tiles = (long long)((p.M + (HM ? 127 : 255)) / (HM ? 128 : 256)) * ((p.N + 255) / 256) * (p.batch > 0 ? p.batch : 1);
        const bool few = (tiles + wg - 1) / wg <= xrounds;
        p.xpf_kt = (few && nk >= xmin && xd > 0) ? (nk - xd > 0 ? nk - xd : 0) : -1;
    }
    hipLaunchKernelGGL(kern, dim3(num_workgroups()), dim3(512), P_LDS, st, p, g_timing);
    return hipGetLastError();
}

template <typename T>
static hipError_t dispatch8(const GemmArgs& p, int epi, hipStream_t st) {
    if (p.dbg) {  // timing experiments (tools/microbench.py; results wrong): ESMK_EXPERIMENTS builds only
#ifdef ESMK_EXPERIMENTS
        if constexpr (std::is_same<T, _Float16>::value) {
            if (epi == EPI_STORE_T) {
                switch (p.dbg) {
                    case 0x10: return launch8<T, EPI_STORE_T, 1, 0>(p, st);
                    case 0x01: return launch8<T, EPI_STORE_T, 0, 1>(p, st);
                    case 0x02: return launch8<T, EPI_STORE_T, 0, 2>(p, st);
                    case 0x04: return launch8<T, EPI_STORE_T, 0, 4>(p, st);
                    case 0x08: return launch8<T, EPI_STORE_T, 0, 8>(p, st);
                    case 0x06: return launch8<T, EPI_STORE_T, 0, 6>(p, st);
                    case 0x07: return launch8<T, EPI_STORE_T, 0, 7>(p, st);
                    case 0x0e: return launch8<T, EPI_STORE_T, 0, 14>(p, st);
                    case 0x18: return launch8<T, EPI_STORE_T, 1, 8>(p, st);
                    case 0x20: return launch8<T, EPI_STORE_T, 0, 16>(p, st);
                    case 0x40: return launch8<T, EPI_STORE_T, 0, 32>(p, st);
                    case 0x60: return launch8<T, EPI_STORE_T, 0, 64>(p, st);
                    case 0x68: return launch8<T, EPI_STORE_T, 0, 64 + 8>(p, st);
                }
            }
            if (p.dbg == 0x90 && epi == EPI_RESID_F32) return launch8<T, EPI_RESID_F32, 0, 128>(p, st);
        }
#endif
        return hipErrorInvalidValue;
    }
    // ESMK_GEMM8_MODE (read once): engine-level A/B of kernel variants with bench.py
    static const int mode = [] {
        const char* e = getenv("ESMK_GEMM8_MODE");
        if (e == nullptr) return 0;
        if (!strcmp(e, "pf4")) return 2;
        if (!strcmp(e, "young")) return 3;
        if (!strcmp(e, "noxpf")) return 4;
        return 0;
    }();
    // generalised addressing requested?  (MSA Transformer calls, batched / strided / remapped GEMMs)
    const bool gen = gemm8_generalised(p, epi);
#define ESMK_CASES(SC, DB, PFD)                                                          \
    switch (epi) {                                                                       \
        case EPI_STORE_T: return launch8<T, EPI_STORE_T, SC, DB, PFD>(p, st);            \
        case EPI_STORE_F32: return launch8<T, EPI_STORE_F32, SC, DB, PFD>(p, st);        \
        case EPI_GELU_T: return launch8<T, EPI_GELU_T, SC, DB, PFD>(p, st);              \
        case EPI_GELU_F32: return launch8<T, EPI_GELU_F32, SC, DB, PFD>(p, st);          \
        case EPI_RESID_F32: return launch8<T, EPI_RESID_F32, SC, DB, PFD>(p, st);        \
        case EPI_QKV_ROPE: return launch8<T, EPI_QKV_ROPE, SC, DB, PFD>(p, st);          \
        case EPI_V_T: return launch8<T, EPI_V_T, SC, DB, PFD>(p, st);                    \
    }
    if (gen) {
        switch (epi) {
            case EPI_STORE_T: return launch8<T, EPI_STORE_T, 0, 0, 0, true>(p, st);
            case EPI_STORE_F32: return launch8<T, EPI_STORE_F32, 0, 0, 0, true>(p, st);
            case EPI_GELU_T: return launch8<T, EPI_GELU_T, 0, 0, 0, true>(p, st);
            case EPI_RESID_F32: return launch8<T, EPI_RESID_F32, 0, 0, 0, true>(p, st);
            case EPI_QKV_ROPE: return launch8<T, EPI_QKV_ROPE, 0, 0, 0, true>(p, st);
            case EPI_V_T: return launch8<T, EPI_V_T, 0, 0, 0, true>(p, st);
            case EPI_MSA_CTX: return launch8<T, EPI_MSA_CTX, 0, 0, 0, true>(p, st);
        }
        return hipErrorInvalidValue;
    }
    if constexpr (std::is_same<T, _Float16>::value) {
        if (mode == 3) { ESMK_CASES(0, 32, 0) }
        if (mode == 4) { ESMK_CASES(0, 128, 0) }
    }
    if (gemm8_half_height(p)) {
        // ESMK_HM_PF=4 (experiment): half-height launches (small batches: latency bound, operands come from the MALL)
        // with the L2 prefetch stream four K tiles ahead
        static const int hm_pf = [] { const char* e = getenv("ESMK_HM_PF"); return e ? atoi(e) : 0; }();
        if (hm_pf == 4) {
            switch (epi) {
                case EPI_STORE_T: return launch8<T, EPI_STORE_T, 0, 0, 4, false, true>(p, st);
                case EPI_GELU_T: return launch8<T, EPI_GELU_T, 0, 0, 4, false, true>(p, st);
                case EPI_RESID_F32: return launch8<T, EPI_RESID_F32, 0, 0, 4, false, true>(p, st);
                case EPI_QKV_ROPE: return launch8<T, EPI_QKV_ROPE, 0, 0, 4, false, true>(p, st);
                case EPI_V_T: return launch8<T, EPI_V_T, 0, 0, 4, false, true>(p, st);
            }
        }
        switch (epi) {
            case EPI_STORE_T: return launch8<T, EPI_STORE_T, 0, 0, 0, false, true>(p, st);
            case EPI_STORE_F32: return launch8<T, EPI_STORE_F32, 0, 0, 0, false, true>(p, st);
            case EPI_GELU_T: return launch8<T, EPI_GELU_T, 0, 0, 0, false, true>(p, st);
            case EPI_GELU_F32: return launch8<T, EPI_GELU_F32, 0, 0, 0, false, true>(p, st);
            case EPI_RESID_F32: return launch8<T, EPI_RESID_F32, 0, 0, 0, false, true>(p, st);
            case EPI_QKV_ROPE: return launch8<T, EPI_QKV_ROPE, 0, 0, 0, false, true>(p, st);
            case EPI_V_T: return launch8<T, EPI_V_T, 0, 0, 0, false, true>(p, st);
        }
    }
    ESMK_CASES(0, 0, PF_DEFAULT)
#undef ESMK_CASES
    return hipErrorInvalidValue;
}

bool gemm8_supports(const GemmArgs& p, int epi) {
    if (p.K % 64 != 0 || p.N % 8 != 0 || p.M <= 0) return false;
    if ((epi == EPI_QKV_ROPE || epi == EPI_V_T || epi == EPI_MSA_CTX) && p.N % 64 != 0) return false;
    return true;
}

hipError_t launch_gemm8(const GemmArgs& p, int epi, int operand_dtype, hipStream_t st) {
    if (!gemm8_supports(p, epi)) return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_F16) return dispatch8<_Float16>(p, epi, st);
    if (operand_dtype == ESMK_DT_BF16) return dispatch8<__bf16>(p, epi, st);
    return hipErrorInvalidValue;
}

}  // namespace esmk
