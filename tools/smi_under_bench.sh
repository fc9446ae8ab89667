#!/bin/bash
# shader clock and package power, sampled by rocm-smi every 0.5 s, while bench.py runs its forward steps.
# usage (on the GPU box): bash tools/smi_under_bench.sh [bench.py arguments...]  -> gpurun_out/smi_under_bench.log
mkdir -p gpurun_out
OUT=gpurun_out/smi_under_bench.log
: > $OUT
( while true; do rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $OUT; echo >> $OUT; sleep 0.5; done ) &
SMI=$!
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary "$@" > gpurun_out/smi_bench.log 2>&1
kill $SMI 2>/dev/null
grep '^{' gpurun_out/smi_bench.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('bench:', r['value'], 'residues/s', r['ms_per_step'], 'ms/step')"
python - <<'PY'
import re
rows = []
for ln in open("gpurun_out/smi_under_bench.log"):
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", ln) or re.search(r"\((\d+)Mhz\)", ln)
    p = re.search(r"Power \(W\): ([\d.]+)", ln) or re.search(r"([\d.]+)\s*W", ln)
    if c and p:
        rows.append((int(c.group(1)), float(p.group(1))))
busy = [r for r in rows if r[1] > 1000]
if busy:
    print(f"{len(busy)} samples above 1000 W: sclk {sum(r[0] for r in busy) / len(busy):.0f} MHz (min {min(r[0] for r in busy)}, max {max(r[0] for r in busy)}), "
          f"power {sum(r[1] for r in busy) / len(busy):.0f} W (max {max(r[1] for r in busy):.0f})")
else:
    print("no loaded samples parsed; raw tail:")
    print("".join(open("gpurun_out/smi_under_bench.log").readlines()[-5:]))
PY
