mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 gpurun_out/r3g/pytest_all.log
grep -E "dims f16|\[f16x2\]|logits rel" gpurun_out/r3g/pytest_all.log | tail
ESMK_GEMM_IMPL=9 timeout 600 python -m pytest tests/test_varlen_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r3g/pytest_impl9.log 2>&1; echo "impl9 tests rc=$?"; tail -3 gpurun_out/r3g/pytest_impl9.log
run() { # tag env args
  env $2 timeout 300 python bench.py $3 --no-cpu-baseline --no-secondary > gpurun_out/r3g/$1.log 2>&1; grep '^{' gpurun_out/r3g/$1.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernel_classes'].items() if 'gemm' in k or 'attention' in k or 'lm_' in k})"
}
run auto "A=1" ""
run impl8 "ESMK_GEMM_IMPL=8" ""
run auto2 "A=1" ""
timeout 300 python bench.py --operand f16x2 > gpurun_out/r3g/f16x2.log 2>&1; grep '^{' gpurun_out/r3g/f16x2.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('f16x2', r['value'], r['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernel_classes'].items() if 'lm_' in k}, r.get('parity'))"
timeout 300 python tools/bench_extract_hosts.py --world 8 --seqs-per-rank 256 --gpu-ms 100 > gpurun_out/r3g/hosts8.log 2>&1; grep '^{' gpurun_out/r3g/hosts8.log
timeout 300 python tools/bench_extract_hosts.py --world 8 --seqs-per-rank 256 --gpu-ms 0 > gpurun_out/r3g/hosts8_nogpu.log 2>&1; grep '^{' gpurun_out/r3g/hosts8_nogpu.log
timeout 300 python tools/bench_extract_hosts.py --world 1 --seqs-per-rank 256 --gpu-ms 0 > gpurun_out/r3g/hosts1_nogpu.log 2>&1; grep '^{' gpurun_out/r3g/hosts1_nogpu.log
