# round 4, call 2: LayerNorm fold — single-op tests, model tests, forward A/B
O=gpurun_out/r4b
mkdir -p $O
timeout 600 python -m pytest tests/test_ln_fold_gpu.py -q -s -x > $O/pytest_fold.log 2>&1; echo "fold tests rc=$?"; tail -25 $O/pytest_fold.log
timeout 600 python -m pytest tests/test_ln_fold_gpu.py -q -s > $O/pytest_fold_all.log 2>&1; echo "fold tests (no -x) rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_fold_all.log | tail -20
for f in 0 1; do
  ESM_AMD_LN_FOLD=$f timeout 300 python bench.py --no-secondary --steps 12 --warmup 4 > $O/bench_fold_$f.log 2>&1
  python - <<PY
import json
for l in open("$O/bench_fold_$f.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print("fold $f:", r["value"], r["ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in r["kernel_classes"].items()})
        print("   parity", {k: v for k, v in (r.get("parity") or {}).items() if not isinstance(v, dict)})
PY
  tail -3 $O/bench_fold_$f.log | cut -c1-300
done
ESM_AMD_LN_FOLD=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline --batch 4 --steps 20 --warmup 5 > $O/bench_fold_1_b4.log 2>&1; grep '^{' $O/bench_fold_1_b4.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('fold 1 B=4', r['value'], r['ms_per_step'])"
ESM_AMD_LN_FOLD=0 timeout 300 python bench.py --no-secondary --no-cpu-baseline --batch 4 --steps 20 --warmup 5 > $O/bench_fold_0_b4.log 2>&1; grep '^{' $O/bench_fold_0_b4.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('fold 0 B=4', r['value'], r['ms_per_step'])"
ESM_AMD_LN_FOLD=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_varlen_gpu.py tests/test_esm1b_gpu.py tests/test_round3_gpu.py tests/test_round2_gpu.py -q -x > $O/pytest_models_fold.log 2>&1; echo "model tests under fold rc=$?"; tail -15 $O/pytest_models_fold.log
