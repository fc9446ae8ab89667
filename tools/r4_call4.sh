# round 4, call 4: LayerNorm fold with full-line h16 stores; attention start-up stagger sweep
O=gpurun_out/r4d
mkdir -p $O
timeout 600 python -m pytest tests/test_ln_fold_gpu.py -q -s -x > $O/pytest_fold.log 2>&1; echo "fold tests rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_fold.log | tail -5
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 12 --warmup 4 > $O/bench_$tag.log 2>&1
  python - <<PY
import json
for l in open("$O/bench_$tag.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print("$tag:", r["value"], r["ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in r["kernel_classes"].items()})
PY
}
run fold0 ESM_AMD_LN_FOLD=0
run fold1 ESM_AMD_LN_FOLD=1
run stag150 ESM_AMD_LN_FOLD=0 ESMK_ATTN_STAGGER=150
run stag300 ESM_AMD_LN_FOLD=0 ESMK_ATTN_STAGGER=300
run stag450 ESM_AMD_LN_FOLD=0 ESMK_ATTN_STAGGER=450
run stag700 ESM_AMD_LN_FOLD=0 ESMK_ATTN_STAGGER=700
run fold0b ESM_AMD_LN_FOLD=0
run fold1b ESM_AMD_LN_FOLD=1
timeout 300 python tools/bench_gemm9.py --no-vendor --no-check --cases resid --rounds 3 --iters 10 > $O/gemm9_resid.log 2>&1; cat $O/gemm9_resid.log | tail -6
