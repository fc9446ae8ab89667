#!/bin/bash
# Round-5 GPU call 1 (one box, so every comparison is same-box): the whole -m gpu suite on the new default (LayerNorm fold
# on, first-K-tile fix, degree-8 GELU for fp16 outputs, full-height one-launch q/k/v), then A/B bench lines against the
# round-4 library (esm_amd/lib/libesmk_prev.so) and against the new library built without the first-K-tile fix
# (esm_amd/lib/variants/libesmk_notie.so).  Output: gpurun_out/r5b1/
set -u
O=gpurun_out/r5b1
mkdir -p $O
LIB=esm_amd/lib/libesmk.so
cp $LIB /tmp/libesmk_new.so
line() {  # $1 = tag, rest = bench args (env through the caller)
  tag=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/$tag.log 2>&1
  grep '^{' $O/$tag.log > $O/$tag.json
  python - "$tag" "$O/$tag.json" <<'EOF'
import sys, json
try:
    r = json.loads(open(sys.argv[2]).read())
    kc = {k: v['ms_per_step'] for k, v in r.get('kernel_classes', {}).items() if v['ms_per_step'] > 0.3}
    print(sys.argv[1], r['value'], r['ms_per_step'], 'ms', r['config'].get('ln_fold'), r.get('library', {}).get('src_hash'), kc, flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e, flush=True)
EOF
}
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s"; tail -5 $O/pytest_gpu.log
grep -E "^contract |FAILED|Error" $O/pytest_gpu.log > $O/contract_lines.txt
line new_b64
cp esm_amd/lib/libesmk_prev.so $LIB
line r4_plain_b64
ESM_AMD_LN_FOLD=1 line r4_fold_b64
ESM_AMD_LN_FOLD=1 line r4_fold_b4 --batch 4 --steps 20 --warmup 5
cp esm_amd/lib/variants/libesmk_notie.so $LIB
line notie_b64
cp /tmp/libesmk_new.so $LIB
line new_plain_b64 --ln-fold 0
line new2_b64
ESMK_QKV_ONE_LAUNCH=1 line new_onelaunch_b64
line new_b16 --batch 16
ESMK_QKV_ONE_LAUNCH=0 line new_b16_twolaunch --batch 16
line new_b4 --batch 4 --steps 20 --warmup 5
line new_3b --workload esm2_3b_contacts --steps 4
cp esm_amd/lib/libesmk_prev.so $LIB
line r4_3b --workload esm2_3b_contacts --steps 4
cp /tmp/libesmk_new.so $LIB
echo "A/B done $(( $(date +%s) - T0 )) s"
T1=$(date +%s); timeout 400 python bench.py > $O/default.log 2>&1; echo "default bench rc=$? wall $(( $(date +%s) - T1 )) s"
grep '^{' $O/default.log > $O/default.json
python - <<'EOF'
import json
r = json.load(open('gpurun_out/r5b1/default.json'))
print('default', r['value'], r['ms_per_step'], r['config'].get('ln_fold'), r['roofline'], r.get('parity'))
for k, v in r.get('secondary_workloads', {}).items():
    print(' ', k, {x: v.get(x) for x in ('value', 'ms_per_step', 'wall_s', 'error', 'skipped')}, v.get('parity'))
EOF
echo "total $(( $(date +%s) - T0 )) s"
