// gemm32.hip — fp32 nn.Linear on the matrix cores, for the LM head of the f16x2 precision mode:
//     out[M,N] = act(A[M,K] . W[N,K]^T + bias)          fp32 in, fp32 accumulate (v_mfma_f32_32x32x2_f32), fp32 out
//
// Replaces RobertaLMHead's two linears (reference esm/modules.py:298-314: dense + GELU, then the tied output
// projection + bias) when the engine runs with split weights: the head is one E x E and one V x E GEMM per forward —
// 220 GFLOP at the 650M size, ~3 ms on the exact-fp32 MFMA path (1/16 of the fp16 rate) against 170 ms for the layer
// stack — and running it in fp32 takes both its weight rounding and its activation rounding out of the logits.
// Plain tile kernel: 128 x 64 output tile per 256-thread workgroup, K step 32 staged through padded LDS rows
// (33 floats: the one-float fragment reads of 32 consecutive rows hit 32 different banks), wave w owns rows
// [32 w, 32 w + 32) x 64 columns.  The MFMA's A operand carries the activation rows, so a lane holds column
// n = lane & 31 of four-row groups and every store instruction writes 128-byte row segments.
#include "common.h"
#include "kernels.h"

namespace esmk {

constexpr int G32_TM = 128, G32_TN = 64, G32_TK = 32, G32_LD = 33;

template <bool GELU>
__global__ __launch_bounds__(256) void gemm32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ out, int ldc,
                                                      int M, int N, int K) {
    __shared__ float As[G32_TM * G32_LD];
    __shared__ float Ws[G32_TN * G32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (N + G32_TN - 1) / G32_TN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * G32_TM, n0 = tn * G32_TN;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int lr = tid >> 3, lc = (tid & 7) * 4;  // staging: row lr (+ 32 it), columns lc .. lc + 3
    for (int k0 = 0; k0 < K; k0 += G32_TK) {
        f32x4 va[4], vw[2];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = min(m0 + lr + 32 * it, M - 1);
            va[it] = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + k0 + lc);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int n = min(n0 + lr + 32 * it, N - 1);
            vw[it] = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k0 + lc);
        }
        __syncthreads();  // the previous K step's fragment reads are done
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(lr + 32 * it) * G32_LD + lc + e] = va[it][e];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[(lr + 32 * it) * G32_LD + lc + e] = vw[it][e];
        __syncthreads();
        const float* ap = As + (32 * wave + (lane & 31)) * G32_LD + (lane >> 5);
        const float* wp = Ws + (lane & 31) * G32_LD + (lane >> 5);
#pragma unroll
        for (int k = 0; k < G32_TK; k += 2) {
            const float a = ap[k];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[32 * G32_LD + k], acc[1], 0, 0, 0);
        }
    }
    // D[m][n]: lane holds column n = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * j + (lane & 31);
        if (n >= N) continue;
        const float b = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * wave + mfma32_row(r, lane >> 5);
            if (m >= M) continue;
            float v = acc[j][r] + b;
            if constexpr (GELU) v = gelu_erf(v);
            out[(size_t)m * ldc + n] = v;
        }
    }
}

hipError_t launch_gemm32(const float* A, int lda, const float* W, const float* bias, float* out, int ldc, int M, int N,
                         int K, bool gelu, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || K % G32_TK != 0 || lda % 4 != 0) return hipErrorInvalidValue;
    const long long blocks = (long long)((M + G32_TM - 1) / G32_TM) * ((N + G32_TN - 1) / G32_TN);
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (gelu) hipLaunchKernelGGL(gemm32_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, A, lda, W, bias, out, ldc, M, N, K);
    else hipLaunchKernelGGL(gemm32_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, A, lda, W, bias, out, ldc, M, N, K);
    return hipGetLastError();
}

}  // namespace esmk
