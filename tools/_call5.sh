mkdir -p gpurun_out/r3h
timeout 240 python tools/bench_gemm9.py --check-only > gpurun_out/r3h/check.log 2>&1; rc=$?; echo "check rc=$rc"; tail -3 gpurun_out/r3h/check.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "check crashed/hung; stopping"; exit 0; fi
timeout 300 python tools/bench_gemm9.py --no-check > gpurun_out/r3h/bench.log 2>&1; echo "bench rc=$?"; cat gpurun_out/r3h/bench.log
ESMK_GEMM_IMPL=9 timeout 300 python -m pytest tests/test_varlen_gpu.py -m gpu -x -q > gpurun_out/r3h/pytest_impl9.log 2>&1; echo "impl9 varlen rc=$?"; tail -2 gpurun_out/r3h/pytest_impl9.log
