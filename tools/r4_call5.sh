O=gpurun_out/r4e
mkdir -p $O
timeout 400 python tools/bench_gemm9.py --no-vendor --no-check --cases "resid" --rounds 3 --iters 10 > $O/gemm9_merged.log 2>&1; grep -v "^device\|subnormal" $O/gemm9_merged.log
timeout 300 python -m pytest tests/test_ln_fold_gpu.py -q -x 2>&1 | tail -3
