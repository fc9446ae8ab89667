O=gpurun_out/r3B
mkdir -p $O
timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_kernels_gpu.py tests/test_round3_gpu.py -q -x 2>&1 | tail -4
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $O/x2_$tag.log 2>&1; python -c "import json; r=json.loads([l for l in open('$O/x2_$tag.log') if l.startswith('{')][-1]); print('$tag', r['value'], r['ms_per_step'], {k: v['ms_per_step'] for k, v in r['kernel_classes'].items() if v['ms_per_step'] > 0.5})"; }
ESMK_GEMM_IMPL=8 run f16x2_gemm8 --operand f16x2
run f16x2_gemm9 --operand f16x2
run f16_650m
