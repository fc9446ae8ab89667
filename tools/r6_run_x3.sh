#!/bin/bash
# Round-6: the f16x3 precision mode (weights and GEMM inputs split): tests, rates
set -u
O=gpurun_out/r6x3
mkdir -p $O
T0=$(date +%s)
timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s -p no:cacheprovider -k "f16x3 or attention_split or split_weight" > $O/pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s"; tail -4 $O/pytest.log; grep -E "contract |esm2_3b_T258" $O/pytest.log | cut -c1-400
timeout 400 python bench.py --steps 6 --warmup 2 --no-secondary --quick-baseline --operand f16x3 > $O/bench_650m_f16x3.json 2> $O/bench_650m_f16x3.err; echo "bench 650m rc=$? $(( $(date +%s) - T0 )) s"
timeout 400 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $O/bench_650m_f16.json 2> $O/bench_650m_f16.err
timeout 400 python bench.py --workload esm2_3b_contacts --steps 3 --warmup 1 --quick-baseline --operand f16x3 > $O/bench_3b_f16x3.json 2> $O/bench_3b_f16x3.err; echo "bench 3b rc=$? $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6x3/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:]); continue
    p = d.get("parity", {})
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["dtype"], {k: (round(v, 6) if isinstance(v, float) else v) for k, v in p.items() if k != "operand_floor_same_inputs"})
    if "operand_floor_same_inputs" in p: print("    floor", p["operand_floor_same_inputs"])
    if "kernel_classes" in d: print("    classes", {k: v.get("ms_per_step") for k, v in d["kernel_classes"].items()})
PY
