mkdir -p gpurun_out/r3e
timeout 240 python tools/bench_gemm9.py --check-only > gpurun_out/r3e/check.log 2>&1; rc=$?; echo "check rc=$rc"; tail -8 gpurun_out/r3e/check.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "check crashed/hung; stopping"; exit 0; fi
timeout 300 python tools/bench_gemm9.py --no-check --dbg > gpurun_out/r3e/bench.log 2>&1; echo "bench rc=$?"; cat gpurun_out/r3e/bench.log
