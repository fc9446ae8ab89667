#!/bin/bash
# Round-5 GPU call 2 (one box): bit-equality tests of the two new kernels forms (attention with 64 query rows per wave,
# residual added in the L2), then same-box A/B bench lines.  Output: gpurun_out/r5b2/
set -u
O=gpurun_out/r5b2
mkdir -p $O
line() {  # $1 = tag, rest = bench args (env through the caller)
  tag=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/$tag.log 2>&1
  grep '^{' $O/$tag.log > $O/$tag.json
  python - "$tag" "$O/$tag.json" <<'PY'
import sys, json
try:
    r = json.loads(open(sys.argv[2]).read())
    kc = {k: v['ms_per_step'] for k, v in r.get('kernel_classes', {}).items() if v['ms_per_step'] > 0.3}
    print(sys.argv[1], r['value'], r['ms_per_step'], 'ms', r['config'].get('ln_fold'), r.get('library', {}).get('src_hash'), kc, flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e, flush=True)
PY
}
T0=$(date +%s)
timeout 600 python -m pytest tests/test_attention_w64_gpu.py tests/test_resid_atomic_gpu.py tests/test_qkv_one_launch_gpu.py "tests/test_model_gpu.py::test_degenerate_lengths" -m gpu -q -x -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s"; tail -8 $O/pytest_new.log
line fold_b64
ESMK_ATTN_W64=1 line fold_w64_b64
line plain_b64 --ln-fold 0
ESMK_RESID_ATOMIC=1 line plain_atomic_b64 --ln-fold 0
ESMK_RESID_ATOMIC=1 ESMK_ATTN_W64=1 line plain_atomic_w64_b64 --ln-fold 0
line fold2_b64
line fold_b4 --batch 4 --steps 20 --warmup 5
ESMK_ATTN_W64=1 line fold_w64_b4 --batch 4 --steps 20 --warmup 5
ESMK_RESID_ATOMIC=1 line plain_atomic_b4 --batch 4 --steps 20 --warmup 5 --ln-fold 0
line fold_b16 --batch 16
ESMK_RESID_ATOMIC=1 line plain_atomic_b16 --batch 16 --ln-fold 0
line fold_3b --workload esm2_3b_contacts --steps 4
ESMK_RESID_ATOMIC=1 ESMK_ATTN_W64=1 line plain_atomic_w64_3b --workload esm2_3b_contacts --steps 4 --ln-fold 0
echo "total $(( $(date +%s) - T0 )) s"
