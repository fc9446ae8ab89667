#!/bin/bash
# Round-6 GPU call: the outlier-channel stress set — the parity tests, the gain check, and the study over magnitudes and modes
set -u
O=gpurun_out/r6s
mkdir -p $O
T0=$(date +%s)
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ln_fold_gpu.py -m gpu -q -s -p no:cacheprovider -k "outlier or gain_check" > $O/pytest_stress.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_stress.log; grep -E "contract |outliers" $O/pytest_stress.log
ESM_AMD_LN_FOLD=0 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -p no:cacheprovider -k "outlier" > $O/pytest_stress_plain.log 2>&1
echo "pytest plain pass rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_stress_plain.log
timeout 1200 python tools/outlier_stress_study.py > $O/outlier_stress_study.log 2> $O/outlier_stress_study.err; echo "study rc=$? $(( $(date +%s) - T0 )) s"
timeout 600 python tools/outlier_stress_study.py --balanced --magnitudes 2000 > $O/outlier_stress_study_balanced.log 2>> $O/outlier_stress_study.err; echo "study balanced rc=$? $(( $(date +%s) - T0 )) s"
grep -A8 "^==" $O/outlier_stress_study.log $O/outlier_stress_study_balanced.log | cut -c1-330
tail -5 $O/outlier_stress_study.err
timeout 300 python bench.py --steps 20 --warmup 5 --batch 4 --no-secondary --quick-baseline > $O/bench_b4.json 2> $O/bench_b4.err; echo "bench b4 rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6s/bench_b4.json").read().strip().splitlines()[-1])
print("B=4", d["value"], d["ms_per_step"], d.get("ln_fold"), d.get("config"))
PY
