// gemm.hip — nn.Linear on the gfx950 matrix cores:  C[M,N] = A[M,K] . W[N,K]^T (+ bias, + epilogue)
//
// Replaces every torch.nn.Linear / F.linear call of the ESM-2 layer stack
// (reference esm/multihead_attention.py:256-261,395; esm/modules.py:138-139,309,313) and fuses
// what the reference runs as separate elementwise ops into the epilogue:
//   * bias add,
//   * q scaling (multihead_attention.py:261), rotary embedding (rotary_embedding.py:11-20,63-69)
//     and the head split / transpose (multihead_attention.py:280-284)        -> EPI_QKV_ROPE (q,k), EPI_V_T (v)
//   * exact-erf GELU (modules.py:17-24)                                      -> EPI_GELU_*
//   * residual add into the fp32 stream (modules.py:134,140)                 -> EPI_RESID_F32
//
// One-tile-per-workgroup kernel (gemm256; the reference the persistent kernels are tested against bit for bit):
// 256x256 output tile, K step 64, 8 waves (2 along M x 4 along N, each
// wave 128x64 = 8x4 v_mfma_f32_16x16x32 blocks), operands staged HBM->LDS with
// global_load_lds_dwordx4 into two 64 KiB LDS buffers (one barrier per K step), LDS rows are
// 128 B with the 16-byte chunk index XOR-swizzled by ((row>>1)&7) so that every ds_read_b128
// lane group touches 16 distinct 16-byte slots (conflict free).  Because global_load_lds
// writes lane-linear, the swizzle is applied to the per-lane SOURCE address and again on the
// read side.  Workgroup ids are remapped so each XCD (private L2) owns a contiguous run of
// tiles with the N index fastest: the 256-row activation panel is fetched from HBM once per
// XCD and re-used from L2 by all N tiles.
//
// The MFMA is issued "swapped" (A operand = weight rows, B operand = activation rows) so a
// lane owns 4 consecutive output columns of one output row: epilogue stores are 8 B (f16/bf16)
// or 16 B (fp32) per lane, and the RoPE partner (column + 32 of the same head) lives in the
// same lane and register index two 16-column blocks further (epilogues: gemm_epi.h).
//
// Generic kernel (gemm64): 64x64 tile, K step 32, register staged with row clamping and
// per-element predicated stores; used for shapes the fast kernel does not cover
// (K % 64 != 0, N % 4 != 0 such as the 33-wide vocabulary projection).
#include "gemm_epi.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace esmk {

// --------------------------------------------------------------------------------------------
// fast kernel: 256 x 256 x 64
// --------------------------------------------------------------------------------------------
constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_TILE_BYTES = G_BM * G_BK * 2;  // 32 KiB per operand per stage
constexpr int G_STAGE_BYTES = 2 * G_TILE_BYTES;
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;  // 128 KiB

// DBG != 0 builds timing-experiment variants (tools/microbench.py --only dbg), results are wrong:
// bit 0 no staging in the loop, bit 1 no vmcnt wait / barrier, bit 2 MFMA only (no LDS reads).
template <typename T, int EPI, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (p.N + G_BN - 1) / G_BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * G_BM, n0 = tn * G_BN;

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W);

    // per-lane staging sources: 4 rounds x (A, W); LDS position (row r, slot s) receives global
    // chunk c = s ^ ((r >> 1) & 7) of row r (16-byte chunks of the 128-byte K slab).
    const T* ga[4];
    const T* gw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pos = j * 512 + tid;
        const int r = pos >> 3, s = pos & 7;
        const int c = (DBG & 16) ? s : (s ^ ((r >> 1) & 7));
        const int ra = min(m0 + r, p.M - 1);
        const int rw = min(n0 + r, p.N - 1);
        ga[j] = A + (size_t)ra * p.K + c * 8;
        gw[j] = W + (size_t)rw * p.K + c * 8;
    }

    // One staging "part" = round j of the activation tile + round j of the weight tile (2 LDS-DMA
    // instructions per wave, 16 KiB per workgroup); a K tile is 4 parts.
    auto stage_part = [&](int buf, int kt, int j) {
        char* base = smem + buf * G_STAGE_BYTES;
        const int ko = (DBG & 32) ? 0 : kt * G_BK;  // DBG 32: re-read K slab 0 (all L2 hits)
        glds16(ga[j] + ko, base + (j * 512 + wave * 64) * 16);
        glds16(gw[j] + ko, base + G_TILE_BYTES + (j * 512 + wave * 64) * 16);
    };
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_part(buf, kt, j);
    };

    // fragment read offsets (bytes) inside a tile: 16 x 16 x 32 MFMA blocks, lane l reads row l & 15 of a 16-row
    // block and 16-byte chunk 4 kh + (l >> 4) of its 128-byte row (kh = K half of the tile)
    const int lrow = (lane & 15) * 128;
    const int swz = (lane >> 1) & 7;
    int xo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xo[ks] = ((4 * ks + (lane >> 4)) ^ swz) << 4;
    const int a_off = (wm * 128) * 128 + lrow;                // activation rows of this wave
    const int w_off = G_TILE_BYTES + (wn * 64) * 128 + lrow;  // weight rows of this wave

    // acc = bias (gemm8's order: the bias rides through the K loop), EPI_V_T adds it in its epilogue
    f32x4 acc[4][8];  // [16-column block][16-row block]
    {
        const int nb = n0 + wn * 64;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + 16 * nj + 4 * (lane >> 4) + r;
                const float b = (EPI != EPI_V_T && p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) acc[nj][mi][r] = b;
            }
    }

    using V8 = typename Op<T>::v8;
    struct Frags {  // one K half: 4 column blocks, 4 of the 8 row blocks (hm = 0: rows 0..63, 1: rows 64..127)
        V8 w[4], a[4];
    };
    // step ks = 0..3: K half ks >> 1, row blocks 4 (ks & 1) .. + 3
    auto read_frags = [&](Frags& f, const char* sb, int ks) {
        const int kh = ks >> 1, hm = ks & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.w[j] = *reinterpret_cast<const V8*>(sb + w_off + j * 2048 + xo[kh]);
#pragma unroll
        for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const V8*>(sb + a_off + (4 * hm + i) * 2048 + xo[kh]);
    };
    auto mma8 = [&](const Frags& f, int ks) {
        const int hm = ks & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4& c = acc[j][4 * hm + i];
                if constexpr (EPI == EPI_V_T)  // lane owns 4 consecutive tokens of one channel
                    c = Op<T>::mma16(f.a[i], f.w[j], c);
                else  // lane owns 4 consecutive channels of one token
                    c = Op<T>::mma16(f.w[j], f.a[i], c);
            }
    };

    // Software pipeline: the fragments of k-slice s+1 are requested from the LDS BEFORE the 8 MFMAs
    // of slice s are issued (two register sets), and the first slice of the NEXT tile is requested
    // right after the tile barrier, in front of the last 8 MFMAs of the current tile, so LDS
    // latency and the barrier skew hide behind matrix work instead of idling the pipe.
    const int nk = p.K / G_BK;
    Frags f0, f1;
    stage(0, 0);
    wait_vmcnt0();
    __syncthreads();
    read_frags(f0, smem, 0);

    // `arrived(f)` is an empty asm that "uses" the fragment registers: hipcc places the LDS wait
    // (lgkmcnt) in front of it, i.e. BEFORE the next slice's reads are issued, when only the needed
    // reads are outstanding (it otherwise waits for the just-issued prefetch as well).
    auto arrived = [&](Frags& f) {
        asm volatile("" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]), "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]));
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk) && !(DBG & 1);
        // The LDS-DMA instructions of the next tile are spread over the MFMA groups (SCHED digits =
        // parts issued in front of group 0..3): a burst of all 64 per CU at the top of the K step
        // parks every wave in the memory-issue queue while the matrix pipe idles.
        constexpr int SCHED = (DBG & 64) ? 0x4000 : ((DBG & 128) ? 0x2200 : 0x2110);
        auto stage_some = [&](int first, int count) {
            if (more) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j >= first && j < first + count) stage_part(cur ^ 1, kt + 1, j);
            }
        };
        constexpr int C0 = (SCHED >> 12) & 15, C1 = (SCHED >> 8) & 15, C2 = (SCHED >> 4) & 15, C3 = SCHED & 15;
        static_assert(C0 + C1 + C2 + C3 == 4, "every part exactly once");
        const char* sb = smem + cur * G_STAGE_BYTES;
        if constexpr (DBG & 8) {  // timing experiment: staging only
            stage_some(0, 4);
            wait_vmcnt0();
            __syncthreads();
            continue;
        }
        if constexpr (DBG & 4) {  // timing experiment: MFMA only
            mma8(f0, 0);
            mma8(f0, 1);
            mma8(f0, 2);
            mma8(f0, 3);
            if constexpr (!(DBG & 2)) __syncthreads();
            continue;
        }
        arrived(f0);
        read_frags(f1, sb, 1);
        stage_some(0, C0);
        __builtin_amdgcn_sched_barrier(0);
        mma8(f0, 0);
        __builtin_amdgcn_sched_barrier(0);
        arrived(f1);
        read_frags(f0, sb, 2);
        stage_some(C0, C1);
        __builtin_amdgcn_sched_barrier(0);
        mma8(f1, 1);
        __builtin_amdgcn_sched_barrier(0);
        arrived(f0);
        read_frags(f1, sb, 3);
        stage_some(C0 + C1, C2);
        __builtin_amdgcn_sched_barrier(0);
        mma8(f0, 2);
        __builtin_amdgcn_sched_barrier(0);
        arrived(f1);
        stage_some(C0 + C1 + C2, C3);
        if constexpr (!(DBG & 2)) {
            wait_vmcnt0();     // next tile has landed (issued one whole K step ago)
            __syncthreads();   // ... for every wave, and every wave is done reading this buffer
        }
        read_frags(f0, smem + (cur ^ 1) * G_STAGE_BYTES, 0);  // (harmless stale read after the last tile)
        __builtin_amdgcn_sched_barrier(0);
        mma8(f1, 3);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // all MFMA operands consumed before the LDS is reused by the epilogue

    epilogue8m<T, EPI, false>(p, acc, 0, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * 16384, 0, 0, 0);
}

// --------------------------------------------------------------------------------------------
// generic kernel: 64 x 64 x 32, any M, N; K % 32 == 0
// --------------------------------------------------------------------------------------------
constexpr int S_ROW = 80;  // bytes per LDS row: 32 elements (64 B) + 16 B pad

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm64_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char sA[64 * S_ROW];
    __shared__ __attribute__((aligned(16))) char sW[64 * S_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + 63) / 64;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W);
    using V8 = typename Op<T>::v8;

    const int lr = tid >> 2, lc = tid & 3;  // 64 rows x 4 chunks of 8 elements
    const T* ga = A + (size_t)min(m0 + lr, p.M - 1) * p.K + lc * 8;
    const T* gw = W + (size_t)min(n0 + lr, p.N - 1) * p.K + lc * 8;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nk = p.K / 32;
    for (int kt = 0; kt < nk; ++kt) {
        const V8 va = *reinterpret_cast<const V8*>(ga + kt * 32);
        const V8 vw = *reinterpret_cast<const V8*>(gw + kt * 32);
        __syncthreads();
        *reinterpret_cast<V8*>(sA + lr * S_ROW + lc * 16) = va;
        *reinterpret_cast<V8*>(sW + lr * S_ROW + lc * 16) = vw;
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = (lane & 31) * S_ROW + ks * 32 + (lane >> 5) * 16;
            const V8 wf = *reinterpret_cast<const V8*>(sW + wn * 32 * S_ROW + off);
            const V8 af = *reinterpret_cast<const V8*>(sA + wm * 32 * S_ROW + off);
            acc = Op<T>::mma(wf, af, acc);
        }
    }

    const int h = lane >> 5;
    const int m = m0 + wm * 32 + (lane & 31);
    if (m >= p.M) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + mfma32_row(r, h);
        if (n >= p.N) continue;
        float v = acc[r] + (p.bias ? p.bias[n] : 0.f);
        if constexpr (EPI == EPI_GELU_T || EPI == EPI_GELU_F32) v = gelu_erf(v);
        const size_t o = (size_t)m * p.N + n;
        if constexpr (EPI == EPI_STORE_T || EPI == EPI_GELU_T)
            reinterpret_cast<T*>(p.out)[o] = Op<T>::from(v);
        else if constexpr (EPI == EPI_RESID_F32)
            reinterpret_cast<float*>(p.out)[o] += v;
        else
            reinterpret_cast<float*>(p.out)[o] = v;
    }
}

// --------------------------------------------------------------------------------------------
// host launchers
// --------------------------------------------------------------------------------------------
template <typename T, int EPI, int DBG = 0>
static hipError_t launch_fast(const GemmArgs& p, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm256_kernel<T, EPI, DBG>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = ((p.M + G_BM - 1) / G_BM) * ((p.N + G_BN - 1) / G_BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), G_LDS_BYTES, st, p);
    return hipGetLastError();
}

template <typename T, int EPI>
static hipError_t launch_generic(const GemmArgs& p, hipStream_t st) {
    const int tiles = ((p.M + 63) / 64) * ((p.N + 63) / 64);
    hipLaunchKernelGGL((gemm64_kernel<T, EPI>), dim3(tiles), dim3(256), 0, st, p);
    return hipGetLastError();
}

template <typename T>
static hipError_t dispatch(const GemmArgs& p, int epi, hipStream_t st) {
    const bool fast = (p.K % G_BK == 0) && (p.N % 8 == 0) && !p.force_generic;
    if (epi == EPI_QKV_ROPE || epi == EPI_V_T) {
        if (!fast || p.N % 64 != 0) return hipErrorInvalidValue;
        return epi == EPI_V_T ? launch_fast<T, EPI_V_T>(p, st) : launch_fast<T, EPI_QKV_ROPE>(p, st);
    }
    if (!fast && (p.K % 32 != 0)) return hipErrorInvalidValue;
#ifdef ESMK_EXPERIMENTS
    if (p.dbg && fast && epi == EPI_STORE_T) {  // timing experiments (results wrong)
        switch (p.dbg) {
            case 1: return launch_fast<T, EPI_STORE_T, 1>(p, st);
            case 2: return launch_fast<T, EPI_STORE_T, 2>(p, st);
            case 3: return launch_fast<T, EPI_STORE_T, 3>(p, st);
            case 5: return launch_fast<T, EPI_STORE_T, 5>(p, st);
            case 7: return launch_fast<T, EPI_STORE_T, 7>(p, st);
            case 8: return launch_fast<T, EPI_STORE_T, 8>(p, st);
            case 16: return launch_fast<T, EPI_STORE_T, 16>(p, st);
            case 24: return launch_fast<T, EPI_STORE_T, 24>(p, st);
            case 40: return launch_fast<T, EPI_STORE_T, 40>(p, st);
            case 32: return launch_fast<T, EPI_STORE_T, 32>(p, st);
            case 64: return launch_fast<T, EPI_STORE_T, 64>(p, st);
            case 128: return launch_fast<T, EPI_STORE_T, 128>(p, st);
            case 96: return launch_fast<T, EPI_STORE_T, 96>(p, st);
            case 160: return launch_fast<T, EPI_STORE_T, 160>(p, st);
        }
    }
#else
    if (p.dbg) return hipErrorInvalidValue;
#endif
#define ESMK_CASE(E)                                               \
    case E:                                                        \
        return fast ? launch_fast<T, E>(p, st) : launch_generic<T, E>(p, st);
    switch (epi) {
        ESMK_CASE(EPI_STORE_T)
        ESMK_CASE(EPI_STORE_F32)
        ESMK_CASE(EPI_GELU_T)
        ESMK_CASE(EPI_GELU_F32)
        ESMK_CASE(EPI_RESID_F32)
    }
#undef ESMK_CASE
    return hipErrorInvalidValue;
}

// Which persistent kernel serves a dense call.  g_impl: 8 = gemm8 always, 9 = gemm9 wherever it applies, 0 = auto.
// Shipped policy (auto, ESMK_GEMM9_POLICY=1): EVERY dense call gemm9 supports goes to gemm9 (mask 127 = all epilogues,
// no minimum K), full- or half-height tiles by rounds over the CUs x tile cost; gemm8 serves the generalised-addressing
// calls.  Both give the same bits.  A/B switches: ESMK_GEMM_IMPL = 8 | 9 | 9:<variant> | auto;  ESMK_GEMM9_MASK = bit
// mask over epilogue codes that may go to gemm9 in auto mode (default 127), ESMK_GEMM9_MIN_K = shortest K of the
// residual GEMM that goes to gemm9 (default 0), ESMK_GEMM9_POLICY=0 = the round-3a rule.  The settings are read from
// the environment ONCE (std::call_once: launches may come from several host threads); esmk_debug_gemm_impl overrides
// the kernel choice only and never suppresses the other variables.
static int g_impl = -1, g_impl_var = 0, g_mask9 = 127, g_mink9 = 0, g_auto_var = 0;
void gemm_set_impl(int impl, int var) {
    g_impl = impl;
    g_impl_var = var;
}

// Start-up delay of one workgroup group in the residual GEMMs (gemm9.hip), as a fraction of a tile's main loop
// (nk K tiles x ~2700 cycles); < 0 = not read yet (ESMK_RESID_DESYNC / ESMK_RESID_DESYNC_GROUP, esmk_debug_set).
// Knobs: atomics, so that a launch thread never reads a half-initialised value; their environment defaults are read inside
// gemm_env_init's call_once like every other ESMK_GEMM_* variable (launches may come from several host threads).
static std::atomic<int> g_lnf_dbg{0};
static std::atomic<int> g_qkv_one{-2};  // -2: not set (environment decides)
static std::atomic<int> g_qkv_one_env{-1};
static std::atomic<double> g_desync{-1.0};
static std::atomic<int> g_desync_group{-1};
constexpr double kDesyncDefault = 0.0;
bool gemm_set_knob(const char* key, double value) {
    if (strcmp(key, "resid_desync") == 0) g_desync = value < 0 ? 0.0 : value;
    else if (strcmp(key, "resid_desync_group") == 0) g_desync_group = (int)value;
    else if (strcmp(key, "lnf_dbg") == 0) {
        if (!kExperiments && (int)value != 0) return false;  // removes parts of the producer epilogue: ESMK_EXPERIMENTS builds only
        g_lnf_dbg = (int)value;
    } else if (strcmp(key, "qkv_one_launch") == 0) g_qkv_one = (int)value < -1 ? -2 : (int)value;
    else return false;
    return true;
}
static void gemm_env_init();
static void desync_for(GemmArgs& q, long long tiles) {
    gemm_env_init();
    // only launches of at least two rounds of tiles: the delay is paid once, a hidden burst is won per further round
    const double ds = g_desync.load();
    if (ds > 0 && tiles >= 512) {
        q.desync = (int)(ds * (double)(q.K / 64) * 2700.0);
        q.desync_group = g_desync_group.load();
    }
}

// ESMK_GEMM_IMPL / ESMK_GEMM9_* are read from the environment ONCE (launches may come from several host threads)
static void gemm_env_init() {
    static std::once_flag env_once;
    std::call_once(env_once, [] {
        if (g_impl < 0) {  // not set through esmk_debug_gemm_impl
            const char* e = getenv("ESMK_GEMM_IMPL");
            g_impl = (e != nullptr && e[0] == '9') ? 9 : (e != nullptr && e[0] == '8') ? 8 : 0;
            g_impl_var = (e != nullptr && e[0] == '9' && e[1] == ':') ? atoi(e + 2) : 0;
        }
        if (const char* m = getenv("ESMK_GEMM9_MASK")) g_mask9 = atoi(m);
        if (const char* k = getenv("ESMK_GEMM9_MIN_K")) g_mink9 = atoi(k);
        if (const char* v = getenv("ESMK_GEMM9_VAR")) g_auto_var = atoi(v);  // issue pattern of the auto choice (2 | 3: A/B)
        if (g_desync.load() < 0) {  // not set through esmk_debug_set
            const char* e = getenv("ESMK_RESID_DESYNC");
            g_desync = e ? atof(e) : kDesyncDefault;
        }
        if (g_desync_group.load() < 0) {
            const char* e = getenv("ESMK_RESID_DESYNC_GROUP");
            g_desync_group = e ? atoi(e) : 0;
        }
        if (const char* e = getenv("ESMK_QKV_ONE_LAUNCH")) g_qkv_one_env = atoi(e);
    });
}

// Cost of a dense gemm9 launch in the unit the tile-height choice below uses: rounds of tiles over the 256 workgroups, a
// half-height tile counted as 0.58 of a full one (profiles/r3_gemm9_half_height_b4.log).
static double gemm9_round_cost(int M, int N) {
    const long long tn = (N + 255) / 256;
    const long long tiles = (long long)((M + 255) / 256) * tn, tiles_h = (long long)((M + 127) / 128) * tn;
    const double cost_f = (double)((tiles + 255) / 256), cost_h = 0.58 * (double)((tiles_h + 255) / 256);
    return cost_h < 0.92 * cost_f ? cost_h : cost_f;
}

// q / k (N = 2E) and v (N = E) as ONE launch (EPI_QKV_ALL)?  Only when it saves rounds: the two launches each round their
// tile count up to whole rounds of 256 workgroups, the combined launch rounds once (B = 1 x 1022 at E = 1280: 80 + 40
// half-height tiles = two part-filled rounds against one of 120; B = 64: 10 + 5 against 15 rounds — no gain, the two launches stay).  ESMK_QKV_ONE_LAUNCH
// = 0 / 1 forces the choice (A/B runs); the results are bit-identical either way.
bool gemm_qkv_one_launch(const GemmArgs& qk) {
    static const bool env_old = [] { const char* e = getenv("ESMK_GEMM"); return e != nullptr && strcmp(e, "old") == 0; }();
    gemm_env_init();
    const int knob = g_qkv_one.load();
    const int mode = knob >= -1 ? knob : g_qkv_one_env.load();  // esmk_debug_set("qkv_one_launch", -1 | 0 | 1) overrides the environment
    GemmArgs all = qk;
    all.N = 3 * qk.E;
    if (mode == 0 || qk.N != 2 * qk.E || g_impl == 8 || env_old || qk.force_old || qk.force_generic || qk.dbg ||
        !gemm9_supports(all, EPI_QKV_ALL))
        return false;
    if (mode == 1) return true;
    // The combined kernel exists with HALF-height tiles only.  A full-height instantiation holding both K loops was built
    // twice: round 4 (accumulator quads shuffled through VGPRs: 1.7 x the time per tile) and round 5 with the quads pinned
    // to the AGPR file — clean K loops in the ISA report, but on the GPU 26.1 against 20.4 ms per step for q / k / v at
    // B = 64 and 7.12 against 6.25 ms at B = 16 (profiles/r5_qkv_one_launch_full_height.log): removed again.
    const long long tiles_h = (long long)((qk.M + 127) / 128) * ((3 * qk.E + 255) / 256);
    return 0.58 * (double)((tiles_h + 255) / 256) < gemm9_round_cost(qk.M, 2 * qk.E) + gemm9_round_cost(qk.M, qk.E) - 0.25;
}

hipError_t launch_gemm(const GemmArgs& p, int epi, int operand_dtype, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return hipErrorInvalidValue;
    gemm_env_init();
    static const bool env_old = [] {
        const char* e = getenv("ESMK_GEMM");
        return e != nullptr && strcmp(e, "old") == 0;
    }();
    if (p.x3_out) {  // the hi | hi | lo output form of the f16x3 mode exists in gemm9 only (full-height tiles)
        if (!gemm9_supports(p, epi)) return hipErrorInvalidValue;
        GemmArgs q = p;
        q.half_m = 0;
        return launch_gemm9(q, epi, operand_dtype, 0, st);
    }
    // the LayerNorm-fold forms of the epilogues exist in gemm9 only: such a call never takes another kernel
    const bool lnf = gemm9_ln_fold(p, epi);
    if (lnf && !gemm9_supports(p, epi)) return hipErrorInvalidValue;
    // the one-launch q / k / v form exists in gemm9 only as well (the caller asks gemm_qkv_one_launch first)
    if (epi == EPI_QKV_ALL && (!gemm9_supports(p, epi) || p.force_old || p.force_generic || p.dbg)) return hipErrorInvalidValue;
    const bool only9 = lnf || epi == EPI_QKV_ALL;
    if (only9 || (!env_old && !p.force_old && !p.force_generic && !p.dbg && g_impl != 8 && gemm9_supports(p, epi))) {
        static const bool hm9 = [] { const char* e = getenv("ESMK_GEMM9_HM"); return e == nullptr || atoi(e) != 0; }();
        if (!only9 && g_impl == 9 && g_impl_var >= 0) {
            GemmArgs q = p;
            if (epi == EPI_RESID_F32 && g_impl_var == 0 && p.half_m <= 0)
                desync_for(q, (long long)((p.M + 255) / 256) * ((p.N + 255) / 256));
            return launch_gemm9(q, epi, operand_dtype, g_impl_var, st);
        }
        // auto: tile height by rounds over the CUs x cost of a tile (a half-height tile costs ~0.58 of a full one:
        // 1470 against 2400 - 2600 cycles per K tile, profiles/r3_gemm9_half_height_b4.log).  ESMK_GEMM9_POLICY=0: the
        // round-3a rule (gemm9 for >= 256 full tiles or where gemm8 would take half-height tiles, gemm8 otherwise).
        static const int policy = [] { const char* e = getenv("ESMK_GEMM9_POLICY"); return e ? atoi(e) : 1; }();
        const long long tn = (p.N + 255) / 256;
        const long long tiles = (long long)((p.M + 255) / 256) * tn, tiles_h = (long long)((p.M + 127) / 128) * tn;
        if (only9 || (((g_mask9 >> epi) & 1) && (epi != EPI_RESID_F32 || p.K >= g_mink9))) {
            bool half, use9;
            if (policy == 0 && !only9) {
                half = p.half_m > 0 || (p.half_m == 0 && tiles < 256 && gemm8_half_height(p));
                use9 = half ? hm9 : tiles >= 256;
            } else {
                const double wg = 256.0;
                const double cost_f = (double)((tiles + 255) / 256), cost_h = 0.58 * (double)((tiles_h + 255) / 256);
                (void)wg;
                half = p.half_m > 0 || (p.half_m == 0 && cost_h < 0.92 * cost_f) || epi == EPI_QKV_ALL;
                use9 = true;
            }
            if (use9) {
                GemmArgs q = p;
                q.half_m = half ? 1 : 0;
                q.lnf_dbg = g_lnf_dbg.load();
                if (epi == EPI_RESID_F32 && !half && g_auto_var == 0 && !lnf) desync_for(q, tiles);
                return launch_gemm9(q, epi, operand_dtype, (half || only9) ? 0 : g_auto_var, st);
            }
        }
    }
    if (!env_old && !p.force_old && !p.force_generic && gemm8_supports(p, epi))
        return launch_gemm8(p, epi, operand_dtype, st);
    if (gemm8_generalised(p, epi)) return hipErrorInvalidValue;  // the tile kernels below only know dense calls
    if (operand_dtype == ESMK_DT_F16) return dispatch<_Float16>(p, epi, st);
    if (operand_dtype == ESMK_DT_BF16) return dispatch<__bf16>(p, epi, st);
    return hipErrorInvalidValue;
}

}  // namespace esmk
