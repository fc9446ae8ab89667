// dma_probe.hip — what bounds the operand stream of the persistent GEMMs?  Standalone (no library):
//     hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
//
// 256 workgroups x 4 waves; every wave runs gemm9's LDS-DMA stream and nothing else: per "K tile" 16 pieces
// (`buffer_load_dwordx4 ... lds`, 8 rows x 128 bytes each) into a two-buffer LDS ring, waited for one K tile later with
// the youngest 14 left in flight (`s_waitcnt vmcnt(14)`), two workgroup barriers per K tile.  Only the ADDRESS PATTERN
// varies:
//   gemm    the GEMM's own: workgroup -> tile through the XCD-blocked walk, 256 activation rows + 256 weight rows,
//           K offset advancing (M = 65536; N, K given), optional K stagger by tile column
//   hot     every workgroup reads the same 64 KiB per K tile (all L2 hits after the first touch)
//   l2      every workgroup cycles through its own 256 KiB (32 workgroups of an XCD: 8 MiB > L2 -> MALL), 128 KiB (4 MiB),
//           64 KiB (2 MiB: L2 resident)
//   hbm     every workgroup streams its own region of a 2 GiB buffer (no reuse at all)
// Reported: microseconds per K tile (64 KiB per CU) and the aggregate rate.  The GEMM loop needs one K tile per
// ~1.1 us (MFMA time at 1.9 GHz); the stream alone runs ~1.3 us in gemm9's own no-MFMA arm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

struct Args {
    const char* a;      // activations [M, K] fp16
    const char* w;      // weights [N, K] fp16
    int M, N, K;
    int mode;           // 0 gemm, 1 hot, 2 private region of `region` bytes per workgroup, 3 gemm + K stagger by tile column
    long long region;   // mode 2
    int ktiles;         // K tiles per workgroup (modes 1, 2); mode 0/3: tiles per workgroup derived from M, N
    int inflight;       // 14 (gemm9) or 0 (drain)
    int aux;            // cache policy bits of the loads
};

template <int AUX>
__global__ __launch_bounds__(256, 1) void stream_kernel(Args p, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned rb = (unsigned)p.K * 2u;
    const int nk = p.K >> 6;
    // gemm9's schedule
    const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8;
    const int total = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int n_my = (cnt > slot) ? (cnt - slot + nslot - 1) / nslot : 0;
    int panel_c = tiles_n <= 6 ? tiles_n : (tiles_n % 5 == 0 ? 5 : (tiles_n % 4 == 0 ? 4 : (tiles_n % 6 == 0 ? 6 : 5)));
    const int panel_full = tiles_m * panel_c;
    auto tile_coords = [&](int it, int& tmi, int& tni) {
        const int o = start + slot + it * nslot;
        const int pnl = o / panel_full;
        const int rem = o - pnl * panel_full;
        const int w = min(panel_c, tiles_n - pnl * panel_c);
        tmi = __builtin_amdgcn_readfirstlane(rem / w);
        tni = __builtin_amdgcn_readfirstlane(pnl * panel_c + (rem - tmi * w));
    };
    __amdgpu_buffer_rsrc_t d_a, d_w;
    unsigned s_koff = 0;
    int s_kt = 0, s_it = 0;
    const bool gemm = (p.mode == 0 || p.mode == 3);
    const int my_tiles = gemm ? n_my : 1;
    const int my_nk = gemm ? nk : p.ktiles;
    if (my_tiles == 0) return;
    unsigned vo_even, vo_odd, rb8;
    if (gemm) {
        const unsigned vo_base = (unsigned)(64 * wave + (lane >> 3)) * rb;
        vo_even = vo_base + (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) << 4);
        vo_odd = vo_base + (unsigned)(((lane & 7) ^ (4 + ((lane >> 4) & 3))) << 4);
        rb8 = 8u * rb;
    } else {  // contiguous 64 KiB per K tile: wave w, piece q -> bytes [32768 operand + 8192 w + 1024 q, + 1024)
        vo_even = vo_odd = (unsigned)(8192 * wave + 16 * lane);
        rb8 = 1024u;
    }
    auto set_tile = [&](int it) {
        s_it = it;
        s_kt = 0;
        if (gemm) {
            int tmi, tni;
            tile_coords(it, tmi, tni);
            s_koff = (p.mode == 3) ? (unsigned)((tni & 3) * (nk >> 2)) * 128u : 0u;
            d_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + (size_t)tmi * 256 * rb), 0, (int)(256u * rb), 0x00020000);
            d_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)tni * 256 * rb), 0, (int)(256u * rb), 0x00020000);
        } else {
            s_koff = 0;
            const char* base = p.mode == 1 ? p.a : p.a + (size_t)blockIdx.x * (size_t)p.region;
            d_a = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
            d_w = __builtin_amdgcn_make_buffer_rsrc((void*)(base + 32768), 0, 0x7fffffff, 0x00020000);
        }
    };
    auto advance = [&]() {
        if (s_kt + 1 < my_nk) {
            s_kt += 1;
            if (gemm) {
                s_koff += 128u;
                if (p.mode == 3 && s_koff == (unsigned)nk * 128u) s_koff = 0;
            } else if (p.mode == 2) {
                s_koff += 65536u;
                if ((long long)s_koff >= p.region) s_koff = 0;
            }
        } else if (s_it + 1 < my_tiles) {
            set_tile(s_it + 1);
        }
    };
    auto issue1 = [&](int k, int buf) {
        const int q = k >> 1;
        char* dst = smem + buf * 65536 + ((k & 1) ? 32768 : 0) + wave * 8192 + q * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds((k & 1) ? d_w : d_a, (lds_ptr)dst, 16, (q & 1) ? vo_odd : vo_even,
                                                 s_koff + (unsigned)q * rb8, 0, AUX);
    };
    set_tile(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) issue1(k, 0);
    advance();
#pragma unroll
    for (int k = 0; k < 16; ++k) issue1(k, 1);
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    unsigned long long c0 = 0, w0 = 0;
    if (tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    int cur = 0;
    const int steps = my_tiles * my_nk;
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        advance();  // stream at position s + 2
        asm volatile("s_barrier" ::: "memory");  // buffer cur has been read by everyone
#pragma unroll
        for (int k = 0; k < 14; ++k) issue1(k, cur);
        if (p.inflight == 14) asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        issue1(14, cur);
        issue1(15, cur);
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
        stamps[blockIdx.x * 2 + 1] = wall_clock64() - w0;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static void run(const char* name, Args p, unsigned long long* stamps, int reps) {
    auto k = p.aux == 2 ? stream_kernel<2> : (p.aux == 1 ? stream_kernel<1> : (p.aux == 17 ? stream_kernel<17> : stream_kernel<0>));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, 0, p, stamps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, 0, p, stamps);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    long long steps;
    if (p.mode == 0 || p.mode == 3) steps = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256) * (p.K / 64) / 256;
    else steps = p.ktiles;
    const double us = ms * 1e3 / reps / (double)steps;
    printf("%-46s %7.3f us per K tile (64 KiB per CU)   %6.1f GB/s per CU   %5.2f TB/s\n", name, us, 65536.0 / us * 1e-3,
           256.0 * 65536.0 / us * 1e-6);
    fflush(stdout);
}

int main() {
    const size_t big = (size_t)2 << 30;
    char* buf;
    unsigned long long* stamps;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 1, big));
    CK(hipMalloc(&stamps, 512 * 8));
    Args p;
    memset(&p, 0, sizeof(p));
    p.a = buf;
    p.w = buf + ((size_t)1 << 30);
    p.M = 65536;
    p.inflight = 14;
    for (int rep = 0; rep < 2; ++rep) {
        p.mode = 0; p.N = 2560; p.K = 1280; run("gemm pattern  N=2560 K=1280 (q,k)", p, stamps, 5);
        p.mode = 3; run("gemm pattern  N=2560 K=1280, K stagger by tn", p, stamps, 5);
        p.mode = 0; p.N = 1280; p.K = 1280; run("gemm pattern  N=1280 K=1280 (v,out)", p, stamps, 8);
        p.mode = 3; run("gemm pattern  N=1280 K=1280, K stagger by tn", p, stamps, 8);
        p.mode = 0; p.N = 5120; p.K = 1280; run("gemm pattern  N=5120 K=1280 (fc1)", p, stamps, 3);
        p.mode = 3; run("gemm pattern  N=5120 K=1280, K stagger by tn", p, stamps, 3);
        p.mode = 0; p.N = 1280; p.K = 5120; run("gemm pattern  N=1280 K=5120 (fc2)", p, stamps, 3);
        p.mode = 3; run("gemm pattern  N=1280 K=5120, K stagger by tn", p, stamps, 3);
        p.mode = 0; p.N = 1280; p.K = 5120; p.inflight = 0; run("gemm pattern  fc2, queue drained per K tile", p, stamps, 3); p.inflight = 14;
        p.mode = 0; p.N = 5120; p.K = 1280; p.aux = 2; run("gemm pattern  fc1, nt loads", p, stamps, 3);
        p.aux = 1; run("gemm pattern  fc1, sc0 loads", p, stamps, 3);
        p.aux = 17; run("gemm pattern  fc1, sc0 sc1 loads", p, stamps, 3); p.aux = 0;
        p.ktiles = 400;
        p.mode = 1; run("hot: all workgroups read the same 64 KiB", p, stamps, 5);
        p.mode = 2; p.region = 65536; run("private 64 KiB per workgroup (L2 resident)", p, stamps, 5);
        p.region = 131072; run("private 128 KiB per workgroup (4 MiB per XCD)", p, stamps, 5);
        p.region = 262144; run("private 256 KiB per workgroup (8 MiB per XCD)", p, stamps, 5);
        p.region = (long long)4 << 20; run("private 4 MiB per workgroup (1 GiB: HBM/MALL)", p, stamps, 5);
        p.inflight = 0; p.region = 65536; run("private 64 KiB, queue drained per K tile", p, stamps, 5); p.inflight = 14;
    }
    return 0;
}
