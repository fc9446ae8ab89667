#!/bin/bash
# Round-6 GPU call 1: the whole -m gpu suite in both engine modes under the tightened contract + a driver-like bench line.
set -u
O=gpurun_out/r6a
mkdir -p $O
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest fold rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log
grep -E "contract |FAILED|Error" $O/pytest_gpu.log > $O/contract_lines.txt
ESM_AMD_LN_FOLD=0 timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu_plain.log 2>&1; echo "pytest plain rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu_plain.log
grep -E "contract |FAILED|Error" $O/pytest_gpu_plain.log > $O/contract_lines_plain.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_650m.json 2> $O/bench_650m.err; echo "bench rc=$? $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6a/bench_650m.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"])
print("parity", {k: v for k, v in d["parity"].items() if k != "operand_floor_same_inputs"})
print("floor", d["parity"].get("operand_floor_same_inputs"))
for k, v in d.get("secondary_workloads", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"))
PY
