// mfma_power_probe.hip — what does one K tile of a 128 x 128 x 64 wave block cost under the package power cap, by MFMA
// shape?  Standalone (no library): hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/probe && /tmp/probe
//
// 256 workgroups x 4 waves (one per SIMD, 512 registers), every wave holds the operand fragments of one K tile in
// 128 VGPRs (random fp16 data, generated per lane) and the 128 x 128 fp32 block in 256 accumulation registers, and runs
// the MFMAs of that K tile over and over:
//     shape 32:  64 x v_mfma_f32_32x32x16_f16   (this repo's kernels)
//     shape 16: 128 x v_mfma_f32_16x16x32_f16   (the hipBLASLt asm kernel: MT256x256x64_MI16x16x1)
// with and without the 32 ds_read_b128 per K tile that refill the fragments.  Reported: time per K tile, shader cycles
// per K tile (s_memtime), the effective clock (cycles / wall), TFLOP/s of the whole chip.  Same flops, same operand
// bytes: whatever differs is the energy per flop of the instruction shape (the chip sits on its 1400 W cap).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ inline unsigned hash_u32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ inline h8 rand_frag(unsigned seed, bool zero) {
    h8 v;
    for (int e = 0; e < 8; ++e) {
        unsigned h = hash_u32(seed * 8u + e);
        // sum of four uniforms: roughly normal, sigma ~ 1
        float f = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) * (1.f / 128.f) - 3.98f;
        v[e] = zero ? (_Float16)0.f : (_Float16)(f * 0.87f);
    }
    return v;
}

template <int SHAPE, bool LDSRD>
__global__ __launch_bounds__(256, 1) void probe(int iters, int zero, float* sink, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // LDS: 64 KiB of random fragments
    for (int i = tid; i < 4096; i += 256) reinterpret_cast<h8*>(smem)[i] = rand_frag(blockIdx.x * 4096 + i, zero != 0);
    __syncthreads();
    h8 fa[16], fb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        fa[i] = rand_frag((blockIdx.x * 256 + tid) * 64 + i, zero != 0);
        fb[i] = rand_frag((blockIdx.x * 256 + tid) * 64 + 32 + i, zero != 0);
    }
    const unsigned rdaddr = (unsigned)(wave * 16384 + lane * 16);
    unsigned long long c0 = 0, w0 = 0;
    if constexpr (SHAPE == 32) {
        f16v acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        if (tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 64; ++m) {  // K sub-step m >> 4, column block (m >> 2) & 3, row block m & 3
                const int ks = m >> 4, j = (m >> 2) & 3, i = m & 3;
                acc[4 * j + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[4 * ks + j], fa[4 * ks + i], acc[4 * j + i], 0, 0, 0);
                if constexpr (LDSRD) {
                    if (m < 32) {  // refill fragment m (it was last used at least one sub-step ago)
                        const int f = (m + 16) & 31;
                        h8& dst = (f < 16) ? fa[f] : fb[f - 16];
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(rdaddr), "n"((m & 15) * 1024));
                    }
                    if (m == 47) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tid == 0) {
            stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
            stamps[blockIdx.x * 2 + 1] = wall_clock64() - w0;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 12345.678f) sink[blockIdx.x * 256 + tid] = s;
    } else {
        f4v acc[64];
#pragma unroll
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        if (tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 128; ++m) {  // K sub-step m >> 6 (32 wide), column block (m >> 3) & 7, row block m & 7
                const int ks = m >> 6, j = (m >> 3) & 7, i = m & 7;
                acc[8 * j + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[8 * ks + j], fa[8 * ks + i], acc[8 * j + i], 0, 0, 0);
                if constexpr (LDSRD) {
                    if (m < 64 && (m & 1) == 0) {
                        const int f = ((m >> 1) + 16) & 31;
                        h8& dst = (f < 16) ? fa[f] : fb[f - 16];
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(rdaddr), "n"(((m >> 1) & 15) * 1024));
                    }
                    if (m == 95) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tid == 0) {
            stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
            stamps[blockIdx.x * 2 + 1] = wall_clock64() - w0;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        if (s == 12345.678f) sink[blockIdx.x * 256 + tid] = s;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE, bool LDSRD>
static void run(const char* name, int zero, float* sink, unsigned long long* stamps, int iters, int launches) {
    auto k = probe<SHAPE, LDSRD>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, 0, iters, zero, sink, stamps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, 0, iters, zero, sink, stamps);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    const double us_per_tile = ms * 1e3 / launches / iters;
    const double flops = 256.0 * 4 * 2.0 * 128 * 128 * 64;  // per K tile of the whole chip
    printf("%-34s %s  %7.3f us/K-tile  %7.1f cycles/K-tile  clock %.2f GHz  %7.1f TFLOP/s\n", name, zero ? "zeros " : "random",
           us_per_tile, cyc / 256 / iters, cyc / wall * 0.1, flops / us_per_tile * 1e-6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;  // ~40 ms per launch
    const int launches = argc > 2 ? atoi(argv[2]) : 12;
    float* sink; unsigned long long* stamps;
    CK(hipMalloc(&sink, 256 * 256 * 4));
    CK(hipMalloc(&stamps, 512 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        run<32, false>("32x32x16  x 64, registers only", 0, sink, stamps, iters, launches);
        run<16, false>("16x16x32 x 128, registers only", 0, sink, stamps, iters, launches);
        run<32, true>("32x32x16  x 64 + 32 ds_read_b128", 0, sink, stamps, iters, launches);
        run<16, true>("16x16x32 x 128 + 32 ds_read_b128", 0, sink, stamps, iters, launches);
    }
    run<32, false>("32x32x16  x 64, registers only", 1, sink, stamps, iters, launches);
    run<16, false>("16x16x32 x 128, registers only", 1, sink, stamps, iters, launches);
    return 0;
}
