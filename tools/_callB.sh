O=gpurun_out/r3B
mkdir -p $O
timeout 300 python bench.py --workload extract_650m --steps 8 --warmup 2 --quick-baseline > $O/extract_parity.log 2>&1; python -c "import json; r=json.loads([l for l in open('$O/extract_parity.log') if l.startswith('{')][-1]); print(r['value'], r.get('parity'), r.get('cpu_baseline'))" || tail -5 $O/extract_parity.log
T0=$(date +%s); timeout 400 python bench.py > $O/default_run.log 2>&1; echo "default wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r3B/default_run.log") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline"].get("traffic_source"))
for k, v in r["secondary_workloads"].items():
    print(k, v.get("value"), v.get("wall_s"), "parity" in v, "cpu_baseline" in v, (v.get("roofline") or {}).get("traffic"), v.get("error"), v.get("skipped"))
PY
