// gemm.hip — nn.Linear on the gfx950 matrix cores:  C[M,N] = A[M,K] . W[N,K]^T (+ bias, + epilogue)
//
// Replaces every torch.nn.Linear / F.linear call of the ESM-2 layer stack
// (reference esm/multihead_attention.py:256-261,395; esm/modules.py:138-139,309,313) and fuses
// what the reference runs as separate elementwise ops into the epilogue:
//   * bias add,
//   * q scaling (multihead_attention.py:261), rotary embedding (rotary_embedding.py:11-20,63-69)
//     and the head split / transpose (multihead_attention.py:280-284)        -> EPI_QKV_ROPE
//   * exact-erf GELU (modules.py:17-24)                                      -> EPI_GELU_*
//   * residual add into the fp32 stream (modules.py:134,140)                 -> EPI_RESID_F32
//
// Fast kernel (gemm256): 256x256 output tile, K step 64, 8 waves (2 along M x 4 along N, each
// wave 128x64 = 4x2 v_mfma_f32_32x32x16 tiles), operands staged HBM->LDS with
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
// same lane and register index of the neighbouring 32-column tile.
//
// Generic kernel (gemm64): 64x64 tile, K step 32, register staged with row clamping and
// per-element predicated stores; used for shapes the fast kernel does not cover
// (K % 64 != 0, N % 4 != 0 such as the 33-wide vocabulary projection).
#include "common.h"
#include "kernels.h"

namespace esmk {

// --------------------------------------------------------------------------------------------
// epilogue
// --------------------------------------------------------------------------------------------
template <typename T>
ESMK_DEV void store4(T* dst, float a, float b, float c, float d) {
    typename Op<T>::v4 v;
    v[0] = Op<T>::from(a);
    v[1] = Op<T>::from(b);
    v[2] = Op<T>::from(c);
    v[3] = Op<T>::from(d);
    *reinterpret_cast<typename Op<T>::v4*>(dst) = v;
}

// One wave owns a [128 (m)] x [64 (n)] block: acc[j][i][r], j = 32-col tile, i = 32-row tile.
// lane: m = m_base + 32 i + (lane & 31);  n = n_base + 32 j + 8 (r>>2) + 4 (lane>>5) + (r&3).
template <typename T, int EPI>
ESMK_DEV void epilogue_wave(const GemmArgs& p, f32x16 (&acc)[2][4], int m_base, int n_base,
                            int lane) {
    const int h = lane >> 5, lm = lane & 31;
    if constexpr (EPI == EPI_QKV_ROPE) {
        // the wave's 64 columns are exactly one head of q, k or v (needs head_dim == 64)
        if (n_base >= p.N) return;
        const int which = n_base / p.E;            // 0 q, 1 k, 2 v   (wave uniform)
        const int head = (n_base - which * p.E) >> 6;
        T* q_or_k = reinterpret_cast<T*>(which == 0 ? p.q : p.k);
        T* vt = reinterpret_cast<T*>(p.vt);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + 32 * i + lm;
            if (m >= p.M) continue;
            const int b = m / p.T, t = m - b * p.T;
            const size_t bh = (size_t)b * p.H + head;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 8 * g + 4 * h;  // first of 4 consecutive dims in [0,32)
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n_base + d0);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.bias + n_base + 32 + d0);
                float x1[4], x2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x1[e] = acc[0][i][4 * g + e] + b1[e];
                    x2[e] = acc[1][i][4 * g + e] + b2[e];
                }
                if (which == 2) {
                    // V transposed: vt[b][head][dv][Tp], key index permuted inside groups of 16
                    // (4-groups 1 and 2 swapped) so the attention kernel reads 8 keys as 16 B.
                    const int t16 = t & 15;
                    const int tp = (t & ~15) | ((((t16 >> 2) & 1) << 3) | (((t16 >> 3) & 1) << 2) |
                                                (t16 & 3));
                    T* base = vt + (bh * 64) * (size_t)p.Tp + tp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        base[(size_t)(d0 + e) * p.Tp] = Op<T>::from(x1[e]);
                        base[(size_t)(32 + d0 + e) * p.Tp] = Op<T>::from(x2[e]);
                    }
                } else {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos + (size_t)t * 32 + d0);
                    const f32x4 s = *reinterpret_cast<const f32x4*>(p.sin + (size_t)t * 32 + d0);
                    float y1[4], y2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a1 = x1[e], a2 = x2[e];
                        if (which == 0) {  // q *= head_dim^-0.5 before the rotation (mha.py:261)
                            a1 *= p.scaling;
                            a2 *= p.scaling;
                        }
                        // x*cos + rotate_half(x)*sin, rotate_half(x) = cat(-x2, x1)
                        y1[e] = a1 * c[e] - a2 * s[e];
                        y2[e] = a2 * c[e] + a1 * s[e];
                    }
                    T* dst = q_or_k + (bh * p.T + t) * 64;
                    store4<T>(dst + d0, y1[0], y1[1], y1[2], y1[3]);
                    store4<T>(dst + 32 + d0, y2[0], y2[1], y2[2], y2[3]);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n_base + 32 * j + 8 * g + 4 * h;
                if (n >= p.N) continue;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m_base + 32 * i + lm;
                    if (m >= p.M) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[j][i][4 * g + e] + bv[e];
                        if constexpr (EPI == EPI_GELU_T || EPI == EPI_GELU_F32) v[e] = gelu_erf(v[e]);
                    }
                    const size_t o = (size_t)m * p.N + n;
                    if constexpr (EPI == EPI_STORE_T || EPI == EPI_GELU_T) {
                        store4<T>(reinterpret_cast<T*>(p.out) + o, v[0], v[1], v[2], v[3]);
                    } else if constexpr (EPI == EPI_RESID_F32) {
                        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o);
                        f32x4 old = *dst;
                        f32x4 nv = {old[0] + v[0], old[1] + v[1], old[2] + v[2], old[3] + v[3]};
                        *dst = nv;
                    } else {
                        f32x4 nv = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = nv;
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// fast kernel: 256 x 256 x 64
// --------------------------------------------------------------------------------------------
constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_TILE_BYTES = G_BM * G_BK * 2;  // 32 KiB per operand per stage
constexpr int G_STAGE_BYTES = 2 * G_TILE_BYTES;
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;  // 128 KiB

template <typename T, int EPI>
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
        const int c = s ^ ((r >> 1) & 7);
        const int ra = min(m0 + r, p.M - 1);
        const int rw = min(n0 + r, p.N - 1);
        ga[j] = A + (size_t)ra * p.K + c * 8;
        gw[j] = W + (size_t)rw * p.K + c * 8;
    }

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * G_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(ga[j] + kt * G_BK, base + (j * 512 + wave * 64) * 16);
            glds16(gw[j] + kt * G_BK, base + G_TILE_BYTES + (j * 512 + wave * 64) * 16);
        }
    };

    // fragment read offsets (bytes) inside a tile
    const int lrow = (lane & 31) * 128;
    const int swz = (lane >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xo[ks] = ((2 * ks + (lane >> 5)) ^ swz) << 4;
    const int a_off = (wm * 128) * 128 + lrow;                // activation rows of this wave
    const int w_off = G_TILE_BYTES + (wn * 64) * 128 + lrow;  // weight rows of this wave

    f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    const int nk = p.K / G_BK;
    stage(0, 0);
    wait_vmcnt0();
    __syncthreads();

    using V8 = typename Op<T>::v8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * G_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            V8 wf[2], af[4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wf[j] = *reinterpret_cast<const V8*>(sb + w_off + j * 4096 + xo[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const V8*>(sb + a_off + i * 4096 + xo[ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = Op<T>::mma(wf[j], af[i], acc[j][i]);
        }
        wait_vmcnt0();
        __syncthreads();
    }

    epilogue_wave<T, EPI>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
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
template <typename T, int EPI>
static hipError_t launch_fast(const GemmArgs& p, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm256_kernel<T, EPI>;
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
    const bool fast = (p.K % G_BK == 0) && (p.N % 4 == 0) && !p.force_generic;
    if (epi == EPI_QKV_ROPE) {
        if (!fast) return hipErrorInvalidValue;
        return launch_fast<T, EPI_QKV_ROPE>(p, st);
    }
    if (!fast && (p.K % 32 != 0)) return hipErrorInvalidValue;
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

hipError_t launch_gemm(const GemmArgs& p, int epi, int operand_dtype, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return hipErrorInvalidValue;
    if (operand_dtype == ESMK_DT_F16) return dispatch<_Float16>(p, epi, st);
    if (operand_dtype == ESMK_DT_BF16) return dispatch<__bf16>(p, epi, st);
    return hipErrorInvalidValue;
}

}  // namespace esmk
