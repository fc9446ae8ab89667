export ESMK_HIPCC_EXTRA="-DESMK_G9_ALIGN=6"
O=gpurun_out/r4j
mkdir -p $O
timeout 400 python tools/bench_gemm9.py --no-vendor --no-check --rounds 3 --iters 10 > $O/gemm9_align6.log 2>&1; grep -v "^device\|subnormal\|amdgpu" $O/gemm9_align6.log | grep -v "no \|none"
