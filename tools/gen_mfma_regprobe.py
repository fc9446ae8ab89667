"""Generate tools/build/mfma_regprobe.hip: does the REGISTER ASSIGNMENT of an MFMA stream change its speed?

gemm9's main loop ran 2630 - 2790 cycles per K tile in one instantiation and 3110 - 3420 in another whose loop has the SAME
instruction sequence and differs only in register numbers (profiles/r4_ln_fold_ablation.log).  This probe replays the 128
v_mfma_f32_16x16x32_f16 of one K tile with the (accumulator, A, B) register triplets of either loop — nothing else: no LDS,
no DMA — from one wave per SIMD, 256 workgroups, and prints cycles per 128 MFMAs.

    python tools/gen_mfma_regprobe.py fast.txt slow.txt ... > tools/build/mfma_regprobe.hip
each file: 128 lines 'acc a b' (first register of the quad)."""
import sys

pats = {}
for f in sys.argv[1:]:
    name = f.split("/")[-1].split(".")[0]
    pats[name] = [tuple(l.split()) for l in open(f) if l.strip()]  # 'acc a b' = MFMA, 'L d' = ds_read_b128 into v[d:d+3]

print("#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <vector>\n#include <algorithm>")
clob = ", ".join([f'"v{i}"' for i in range(252)] + [f'"a{i}"' for i in range(256)])  # v252..255 stay the compiler's
NL = "\\n\\t"
for name, p in pats.items():
    def ins(t):
        if t[0] == "L":
            d = int(t[1])
            return f"ds_read_b128 v[{d}:{d+3}], v252"
        c, a, b = (int(x) for x in t)
        return f"v_mfma_f32_16x16x32_f16 a[{c}:{c+3}], v[{a}:{a+3}], v[{b}:{b+3}], a[{c}:{c+3}]"
    body = NL.join(ins(t) for t in p) + NL + "s_waitcnt lgkmcnt(0)"
    init = " ".join(('"v_mul_f32 v%d, v%d, v0 ' % (i, i - 1) if i % 7 else '"v_add_f32 v%d, v%d, v0' % (i, i - 1)) + NL + '"' for i in range(1, 252))
    print("__global__ __launch_bounds__(256, 1) void probe_" + name + "(unsigned long long* out, int iters, float seed) {")
    print("    unsigned long long t0, t1;")
    print("    __shared__ float lds[4096]; lds[threadIdx.x] = seed; __syncthreads();")
    print("    const unsigned la = (unsigned)(threadIdx.x & 63) * 16u;")
    print("    asm volatile(")
    print('        "v_mov_b32 v252, %4' + NL + 'v_mov_b32 v0, %3' + NL + '"')
    print("        " + init)
    print('        "s_nop 7' + NL + 's_memtime %0' + NL + 's_waitcnt lgkmcnt(0)' + NL + '"')
    print('        "s_mov_b32 s20, %2' + NL + '"')
    print('        "1:' + NL + '"')
    print('        "' + body + NL + '"')
    print('        "s_sub_u32 s20, s20, 1' + NL + 's_cmp_lg_u32 s20, 0' + NL + 's_cbranch_scc1 1b' + NL + '"')
    print('        "s_nop 7' + NL + 's_nop 7' + NL + 's_memtime %1' + NL + 's_waitcnt lgkmcnt(0)' + NL + '"')
    print('        : "=&s"(t0), "=&s"(t1) : "s"(iters), "s"(seed), "v"(la) : "s20", "scc", "memory", "v252", ' + clob + ");")
    print("    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;")
    print("}")
print("""
int main() {
    unsigned long long* d;
    hipMalloc(&d, 1024 * 8);
    std::vector<unsigned long long> h(1024);
    const int iters = 200;""")
for rep in range(2):
    for name in pats:
        print(f"""    hipLaunchKernelGGL(probe_{name}, dim3(256), dim3(256), 0, 0, d, iters, 1.0001f);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-10s: median %.1f cycles per 128 MFMAs (min %.1f max %.1f)\\n", "{name}", (double)h[512] / iters, (double)h[0] / iters, (double)h[1023] / iters);""")
print("    return 0;\n}")
