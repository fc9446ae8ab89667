// mfma_valu_overlap_probe.hip — do the matrix pipe and the VALU of a gfx950 SIMD overlap, across the resident waves of a SIMD
// and inside one wave?  The question behind the attention kernel's ceiling: per 32 x 64 wave-tile at head_dim 64 it
// needs 16 v_mfma_f32_32x32x16 (512 matrix-pipe cycles) and ~80 VALU instructions (32 v_exp_f32, 32 v_add_f32, 16 v_cvt_pk).
// Standalone (no library, no memory traffic — registers only):
//     hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
//
// Every wave runs `iters` wave-tiles of one of four instruction streams:
//   0  MFMA only      16 MFMAs (two chains of four into the two score blocks, two chains of four into the two output blocks)
//   1  VALU only      32 exp + 32 add + 16 cvt_pk on the score registers
//   2  phases         8 score MFMAs -> the 80 VALU -> 8 output MFMAs        (attn_fwd_kernel's order inside a wave)
//   3  interleaved    the 8 score MFMAs of tile t+1 issued BETWEEN the VALU of tile t (1 MFMA : 5 VALU, sched_group_barrier),
//                     then the 8 output MFMAs with the remaining VALU between them (what a software-pipelined wave would do)
// with 1, 2 or 3 workgroups of 4 waves per CU (one compiled kernel per stream, 168 registers).  Reported per configuration: time and
// shader cycles per wave-tile AND PER SIMD (kernel time / (iters x waves per SIMD)): if the pipes overlap, stream 2 at three waves
// approaches max(stream 0, stream 1); if they add, it stays at their sum.  The package power cap moves the clock between streams,
// so times and cycles are both printed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash_u32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ inline h8 rand_frag(unsigned seed) {
    h8 v;
    for (int e = 0; e < 8; ++e) {
        unsigned h = hash_u32(seed * 8u + e);
        float f = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) * (1.f / 128.f) - 3.98f;
        v[e] = (_Float16)(f * 0.3f);
    }
    return v;
}

// the softmax-like VALU block on one 16-register score block: 16 exp, 16 add, 8 cvt_pk -> two packed fragments
__device__ inline void valu_block(const f16v& s, float& ps, h8& p0, h8& p1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = __builtin_amdgcn_exp2f(s[e]), b = __builtin_amdgcn_exp2f(s[8 + e]);
        ps += a;
        ps += b;
        p0[e] = (_Float16)a;
        p1[e] = (_Float16)b;
    }
}

template <int MODE, int W>
__global__ __launch_bounds__(256, W) void probe(int iters, float* sink, unsigned long long* stamps) {
    const int tid = threadIdx.x;
    // operand fragments: q (B operand of the score MFMAs), k (A operand of every MFMA: the register budget of three waves per
    // SIMD, 168, has to hold stream 3's second score block as well).  Scores start from C = 0 (inline constant).
    h8 q[4], k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = rand_frag((blockIdx.x * 256 + tid) * 32 + i), k[i] = rand_frag((blockIdx.x * 256 + tid) * 32 + 4 + i);
    f16v s[2], o[2];
    const f16v c0v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) s[0][r] = s[1][r] = -1.f - 0.01f * r, o[0][r] = o[1][r] = 0.f;
    h8 p[4], p2[4];
    f16v s2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = p2[i] = q[i];
    s2[0] = s[0], s2[1] = s[1];
    float ps = 0.f;
    unsigned long long c0 = 0, w0 = 0;
    if (tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));  // nothing in the loop is invariant for the compiler
        if constexpr (MODE == 0) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[0], q[t2], c0v, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[ks], q[(ks + t2) & 3], s[t2], 0, 0, 0);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[kk], p[(kk + d) & 3], o[d], 0, 0, 0);
            asm volatile("" : "+v"(s[0]), "+v"(s[1]));
        } else if constexpr (MODE == 1) {
            valu_block(s[0], ps, p[0], p[1]);
            valu_block(s[1], ps, p[2], p[3]);
            asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(s[0]), "+v"(s[1]));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[0], q[t2], c0v, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[ks], q[(ks + t2) & 3], s[t2], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            valu_block(s[0], ps, p[0], p[1]);
            valu_block(s[1], ps, p[2], p[3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[kk], p[(kk + d) & 3], o[d], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // software pipeline, two wave-tiles per loop trip so that no register copies are needed: the VALU of tile t (on sa,
            // into pb) beside the score MFMAs of tile t + 1 (into sb), then the output MFMAs of tile t - 1 (from pa) beside the rest
            auto step = [&](f16v (&sa)[2], f16v (&sb)[2], h8 (&pa)[4], h8 (&pb)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    sb[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[0], q[t2], c0v, 0, 0, 0);
#pragma unroll
                    for (int ks = 1; ks < 4; ++ks) sb[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[ks], q[(ks + t2) & 3], sb[t2], 0, 0, 0);
                }
                valu_block(sa[0], ps, pb[0], pb[1]);
#pragma unroll
                for (int g = 0; g < 8; ++g) {  // 8 MFMAs, 5 VALU behind each
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[kk], pa[(kk + d) & 3], o[d], 0, 0, 0);
                valu_block(sa[1], ps, pb[2], pb[3]);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            step(s, s2, p, p2);
            asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
            step(s2, s, p2, p);
            ++it;  // two wave-tiles per trip
        }
    }
    if (tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
        stamps[blockIdx.x * 2 + 1] = wall_clock64() - w0;
    }
    float acc = ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += s[0][r] + s[1][r] + o[0][r] + o[1][r];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += (float)p[i][0] + (MODE == 3 ? (float)p2[i][0] + s2[0][i] + s2[1][i] : 0.f);
    if (acc == 12345.678f) sink[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
static void run(int resident, int iters, float* sink, unsigned long long* d_st, int cus) {
    // ONE compiled kernel per stream (168-register budget of three waves per SIMD); `resident` workgroups per CU are launched
    const int grid = cus * resident;
    hipLaunchKernelGGL((probe<MODE, 3>), dim3(grid), dim3(256), 0, 0, 8, sink, d_st);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, 3>), dim3(grid), dim3(256), 0, 0, iters, sink, d_st);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> st(grid * 2);
    (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < grid; ++i) cyc += (double)st[2 * i], wall += (double)st[2 * i + 1];
    cyc /= grid;
    wall /= grid;
    const char* names[] = {"MFMA only  ", "VALU only  ", "phases     ", "interleaved"};
    // per SIMD: iters x resident wave-tiles in the kernel's time (the workgroups of a CU need not start together, so the kernel time,
    // not a workgroup's own stamps, is the throughput); the clock is the stamped shader cycles over the stamped wall time
    const double ns = ms * 1e6 / ((double)iters * resident), ghz = cyc / (wall * 10.0);
    printf("stream %d %s  %d wave(s) per SIMD: %7.1f ns = %6.0f cycles per wave-tile and SIMD   (kernel %.3f ms, clock %.2f GHz)\n",
           MODE, names[MODE], resident, ns, ns * ghz, ms, ghz);
}

int main() {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    float* sink;
    unsigned long long* d_st;
    (void)hipMalloc(&sink, sizeof(float) * cus * 3 * 256);
    (void)hipMalloc(&d_st, sizeof(unsigned long long) * cus * 3 * 2);
    const int iters = 4000;
    printf("per wave-tile: 16 x v_mfma_f32_32x32x16_f16 = 512 matrix-pipe cycles; 32 v_exp_f32 + 32 v_add_f32 + 16 v_cvt_pk_f16_f32\n");
    for (int w = 1; w <= 3; ++w) run<0>(w, iters, sink, d_st, cus);
    for (int w = 1; w <= 3; ++w) run<1>(w, iters, sink, d_st, cus);
    for (int w = 1; w <= 3; ++w) run<2>(w, iters, sink, d_st, cus);
    for (int w = 1; w <= 3; ++w) run<3>(w, iters, sink, d_st, cus);
    return 0;
}
