#!/bin/bash
# Round-6 GPU call 3: f16x2v test + rates of the three split modes in one call, the default driver-like line with its new secondaries
set -u
O=gpurun_out/r6c
mkdir -p $O
T0=$(date +%s)
timeout 600 python -m pytest tests/test_f16x2_gpu.py -m gpu -q -s -p no:cacheprovider -k "attention_split" > $O/pytest_f16x2v.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_f16x2v.log; grep -E "contract " $O/pytest_f16x2v.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_650m.json 2> $O/bench_650m.err; echo "bench rc=$? $(( $(date +%s) - T0 )) s"
for op in f16x2v f16x2a; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-secondary --quick-baseline --operand $op > $O/bench_$op.json 2> $O/bench_$op.err; echo "bench $op rc=$? $(( $(date +%s) - T0 )) s"
done
timeout 300 python bench.py --workload esm2_3b_contacts --steps 4 --quick-baseline --operand f16x2a > $O/bench_3b_f16x2a.json 2> $O/bench_3b_f16x2a.err; echo "3b f16x2a rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    p = d.get("parity", {})
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["dtype"], {k: (round(v, 6) if isinstance(v, float) else v) for k, v in p.items() if k != "operand_floor_same_inputs"})
    if "operand_floor_same_inputs" in p: print("    floor", p["operand_floor_same_inputs"])
    for k, v in d.get("secondary_workloads", {}).items():
        print("   ", k, v.get("value"), v.get("ms_per_step"), v.get("wall_s"), v.get("skipped"), v.get("error"), v.get("also_qk_gain"), {kk: (round(vv, 6) if isinstance(vv, float) else vv) for kk, vv in (v.get("parity") or {}).items() if kk != "operand_floor_same_inputs"})
PY
