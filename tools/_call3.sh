mkdir -p gpurun_out/r3f
timeout 600 python -m pytest tests/test_f16x2_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -s -k "split or f16x2" > gpurun_out/r3f/pytest_f16x2.log 2>&1; echo "f16x2 tests rc=$?"; grep -E "passed|failed|error|dims f16|epi|\[f16x2\]|esm2_3b" gpurun_out/r3f/pytest_f16x2.log | tail -30
ESMK_GEMM_IMPL=9 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_varlen_gpu.py -m gpu -x -q > gpurun_out/r3f/pytest_impl9.log 2>&1; echo "impl9 tests rc=$?"; tail -3 gpurun_out/r3f/pytest_impl9.log
run() { # tag env args
  env $2 timeout 300 python bench.py $3 --no-cpu-baseline --no-secondary > gpurun_out/r3f/$1.log 2>&1; grep '^{' gpurun_out/r3f/$1.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernel_classes'].items() if 'gemm' in k or 'attention' in k})"
}
run auto "A=1" ""
run impl8 "ESMK_GEMM_IMPL=8" ""
run impl9 "ESMK_GEMM_IMPL=9" ""
run auto2 "A=1" ""
run mask_fc1fc2 "ESMK_GEMM9_MASK=20 ESMK_GEMM9_MIN_K=1024" ""
run mask_all_resid "ESMK_GEMM9_MASK=16 ESMK_GEMM9_MIN_K=1024" ""
run mask_qk "ESMK_GEMM9_MASK=48 ESMK_GEMM9_MIN_K=1024" ""
timeout 300 python bench.py --operand f16x2 > gpurun_out/r3f/f16x2.log 2>&1; grep '^{' gpurun_out/r3f/f16x2.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('f16x2', r['value'], r['ms_per_step'], r.get('parity'))"
