#!/bin/bash
# Round-5 evidence pass B (one box), same library as pass A: smoke, the driver's own line (python bench.py --gpus 1 --steps 20
# --warmup 5; its roofline.traffic now finds pass A's PMC summaries for this library), per-GPU batch sweep, bf16 / f16x2 operands.
# Output: gpurun_out/r5bb/
set -u
O=gpurun_out/r5bb
mkdir -p $O
T0=$(date +%s)
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T1=$(date +%s); timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_650m.log 2>&1; echo "default bench rc=$? wall $(( $(date +%s) - T1 )) s"
grep '^{' $O/bench_650m.log > $O/bench_650m.json
for spec in "b1:--batch 1 --steps 20 --warmup 5" "b4:--batch 4 --steps 20 --warmup 5" "b8:--batch 8 --steps 20 --warmup 5" "b16:--batch 16" "b32:--batch 32" "bf16:--operand bf16" "f16x2:--operand f16x2"; do
  tag=${spec%%:*}; a=${spec#*:}
  timeout 300 python bench.py $a --no-secondary --no-cpu-baseline > $O/bench_650m_$tag.log 2>&1; grep '^{' $O/bench_650m_$tag.log > $O/bench_650m_$tag.json
  python -c "import json; r=json.load(open('$O/bench_650m_$tag.json')); print('$tag', r['value'], r['ms_per_step'], r['config'].get('ln_fold'))"
done
python - <<'PY'
import json
r = json.load(open('gpurun_out/r5bb/bench_650m.json'))
print('default', r['value'], r['ms_per_step'], r['config'].get('ln_fold'), r['roofline'], r.get('parity'))
for k, v in r.get('secondary_workloads', {}).items():
    print(' ', k, {x: v.get(x) for x in ('value', 'ms_per_step', 'wall_s', 'error', 'skipped')}, (v.get('roofline') or {}).get('traffic'), v.get('parity'))
PY
echo "total $(( $(date +%s) - T0 )) s"
