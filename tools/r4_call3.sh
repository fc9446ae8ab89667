# round 4, call 3: LayerNorm fold after the epilogue fixes + the reference extract script on the engine
O=gpurun_out/r4c
mkdir -p $O
timeout 600 python -m pytest tests/test_ln_fold_gpu.py tests/test_extract_script_gpu.py -q -s > $O/pytest_fold.log 2>&1; echo "fold + extract-script tests rc=$?"; grep -E "passed|failed|FAILED|Error|worst|ran " $O/pytest_fold.log | tail -20
for f in 0 1 0 1; do
  ESM_AMD_LN_FOLD=$f timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 12 --warmup 4 > $O/bench_fold_$f.log 2>&1
  python - <<PY
import json
for l in open("$O/bench_fold_$f.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print("fold $f:", r["value"], r["ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in r["kernel_classes"].items()})
PY
done
for b in 4 16; do for f in 0 1; do
ESM_AMD_LN_FOLD=$f timeout 300 python bench.py --no-secondary --no-cpu-baseline --batch $b --steps 20 --warmup 5 > $O/bench_fold_${f}_b$b.log 2>&1; grep '^{' $O/bench_fold_${f}_b$b.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('fold $f B=$b', r['value'], r['ms_per_step'])"
done; done
ESM_AMD_LN_FOLD=1 timeout 300 python bench.py --workload esm2_3b_contacts --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_3b_fold1.log 2>&1; grep '^{' $O/bench_3b_fold1.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('3B fold 1', r['value'], r['ms_per_step'], {k: round(v['ms_per_step'], 2) for k, v in r['kernel_classes'].items()})"
ESM_AMD_LN_FOLD=0 timeout 300 python bench.py --workload esm2_3b_contacts --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_3b_fold0.log 2>&1; grep '^{' $O/bench_3b_fold0.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('3B fold 0', r['value'], r['ms_per_step'], {k: round(v['ms_per_step'], 2) for k, v in r['kernel_classes'].items()})"
