O=gpurun_out/r4e
mkdir -p $O
timeout 400 python tools/bench_gemm9.py --no-vendor --no-check --cases "resid" --rounds 3 --iters 10 > $O/gemm9_reread.log 2>&1; grep -v "^device\|subnormal\|no \|amdgpu" $O/gemm9_reread.log
timeout 300 python -m pytest tests/test_ln_fold_gpu.py -q -x 2>&1 | tail -2
for spec in "0:64" "1:64" "0:4" "1:4" "0:16" "1:16"; do f=${spec%%:*}; b=${spec#*:}
ESM_AMD_LN_FOLD=$f timeout 300 python bench.py --no-secondary --no-cpu-baseline --batch $b --steps 12 --warmup 4 > $O/bench_f${f}_b$b.log 2>&1; grep '^{' $O/bench_f${f}_b$b.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('fold $f B=$b', r['value'], r['ms_per_step'], {k: round(v['ms_per_step'],2) for k,v in r['kernel_classes'].items() if k in ('layernorm','gemm_out_proj','gemm_fc2','gemm_fc1_gelu','gemm_qkv_rope')})"
done
