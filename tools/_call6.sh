mkdir -p gpurun_out/r3i
timeout 300 python tools/bench_gemm9.py --no-check --dbg --no-vendor > gpurun_out/r3i/bench.log 2>&1; echo "bench rc=$?"; grep -E "gemm8|gemm9 |dense-issue|stagger|b24/52|no-barrier" gpurun_out/r3i/bench.log
timeout 400 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "round3 or asm or composes or pad or choice or chunked" > gpurun_out/r3i/pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3i/pytest.log
