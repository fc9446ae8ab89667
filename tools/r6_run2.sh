#!/bin/bash
# Round-6 GPU call 2: the f16x2a precision mode (tests + bench lines at B = 64), the data-sensitivity line
set -u
O=gpurun_out/r6b
mkdir -p $O
T0=$(date +%s)
timeout 600 python -m pytest tests/test_f16x2_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s -p no:cacheprovider -k "split or f16x2" > $O/pytest_f16x2.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_f16x2.log; grep -E "contract |650M dims" $O/pytest_f16x2.log
for op in f16 f16x2a f16x2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --operand $op --ln-fold $([ $op = f16 ] && echo 1 || echo 0) > $O/bench_$op.json 2> $O/bench_$op.err; echo "bench $op rc=$? $(( $(date +%s) - T0 )) s"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-secondary --quick-baseline --qk-gain 4 --ln-gamma-std 0.1 > $O/bench_sharp.json 2> $O/bench_sharp.err; echo "bench sharp rc=$? $(( $(date +%s) - T0 )) s"
timeout 300 python bench.py --steps 10 --warmup 3 --no-secondary --quick-baseline > $O/bench_default_quick.json 2> $O/bench_default_quick.err
timeout 300 python bench.py --workload msa1b --operand f16x2a --quick-baseline > $O/bench_msa_f16x2a.json 2> $O/bench_msa_f16x2a.err; echo "msa f16x2a rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6b/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    p = d.get("parity", {})
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["dtype"], d["config"].get("ln_fold"), {k: (round(v, 6) if isinstance(v, float) else v) for k, v in p.items() if k != "operand_floor_same_inputs"})
    if "operand_floor_same_inputs" in p: print("    floor", p["operand_floor_same_inputs"])
    if "kernel_classes" in d: print("    classes", {k: v.get("ms") if isinstance(v, dict) else v for k, v in d["kernel_classes"].items()} if isinstance(d["kernel_classes"], dict) else d["kernel_classes"])
PY
