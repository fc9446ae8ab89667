#!/bin/bash
# Round-5 evidence pass C (another box, final bench.py): the driver's own line once more, shader clock / package power under the
# forward steps, vendor calibration (hipBLASLt and SDPA on the same operands in the same process).  Output: gpurun_out/r5c/
set -u
O=gpurun_out/r5c
mkdir -p $O
T0=$(date +%s)
T1=$(date +%s); timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_650m.log 2>&1; echo "default bench rc=$? wall $(( $(date +%s) - T1 )) s"
grep '^{' $O/bench_650m.log > $O/bench_650m.json
timeout 200 bash tools/smi_under_bench.sh > $O/rocm_smi_under_load.log 2>&1; tail -3 $O/rocm_smi_under_load.log
timeout 300 python tools/bench_vendor_gemm.py --smi > $O/vendor_gemm_calibration.log 2>&1; tail -12 $O/vendor_gemm_calibration.log
timeout 200 python tools/bench_vendor_attention.py > $O/vendor_attention.log 2>&1; tail -6 $O/vendor_attention.log
python - <<'PY'
import json
r = json.load(open('gpurun_out/r5c/bench_650m.json'))
print('default', r['value'], r['ms_per_step'], r['config'].get('ln_fold'), r['roofline']['frac'], r['roofline']['traffic'])
for k, v in r.get('secondary_workloads', {}).items():
    print(' ', k, {x: v.get(x) for x in ('value', 'ms_per_step', 'steps', 'wall_s', 'error', 'skipped')})
PY
echo "total $(( $(date +%s) - T0 )) s"
