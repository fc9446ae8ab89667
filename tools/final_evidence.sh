# final evidence of a round (round 4) in ONE gpurun call: tests, smoke, bench lines of every workload, vendor calibration, rocprofv3 kernel stats + PMC passes.  usage: gpurun --timeout 3000 -- bash tools/final_evidence.sh ; outputs under gpurun_out/r4final
O=gpurun_out/r4final
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
grep -E "rel|err|L2|argmax|logit|floor|MSA \(|MSA 4|dims|worst|consumer|folded|fold " $O/pytest_gpu.log | grep -v "^tests/" > $O/gpu_tests_parity_lines.txt
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s); timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_650m.log 2>&1; echo "default bench rc=$? wall $(( $(date +%s) - T0 )) s"
grep '^{' $O/bench_650m.log > $O/bench_650m.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4final/bench_650m.json"))
print("650m", r["value"], r["ms_per_step"], r["roofline"]["frac"], {k: v["ms_per_step"] for k, v in r["kernel_classes"].items()})
print("parity", {k: v for k, v in r.get("parity", {}).items() if not isinstance(v, dict)})
print("cpu", r.get("cpu_baseline", {}).get("value"))
for k, v in r.get("secondary_workloads", {}).items():
    print("secondary", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "wall_s", "error", "skipped")}, "parity" in v, "cpu_baseline" in v, (v.get("roofline") or {}).get("traffic"))
PY
for spec in "b1:--batch 1" "b4:--batch 4" "b4_lnfold:--batch 4 --ln-fold 1" "b8:--batch 8" "b16:--batch 16" "b16_lnfold:--batch 16 --ln-fold 1" "b32:--batch 32" "lnfold:--ln-fold 1" "bf16:--operand bf16" "f16x2:--operand f16x2"; do
  tag=${spec%%:*}; a=${spec#*:}
  timeout 300 python bench.py $a --no-secondary > $O/bench_650m_$tag.log 2>&1; grep '^{' $O/bench_650m_$tag.log > $O/bench_650m_$tag.json
  python -c "import json; r=json.load(open('$O/bench_650m_$tag.json')); print('$tag', r['value'], r['ms_per_step'], (r.get('parity') or {}).get('rel_repr_diff_vs_cpu'), (r.get('parity') or {}).get('logits_rel_diff'))"
done
for wl in esm2_3b_contacts msa1b extract_650m; do
  timeout 400 python bench.py --workload $wl > $O/bench_$wl.log 2>&1; grep '^{' $O/bench_$wl.log > $O/bench_$wl.json
  python -c "import json; r=json.load(open('$O/bench_$wl.json')); print('$wl', r['value'], r['ms_per_step'], r.get('parity'))"
done
timeout 400 python bench.py --workload msa1b --operand f16x2 > $O/bench_msa_f16x2.log 2>&1; grep '^{' $O/bench_msa_f16x2.log > $O/bench_msa_f16x2.json; python -c "import json; r=json.load(open('$O/bench_msa_f16x2.json')); print('msa f16x2', r['value'], r['ms_per_step'], r.get('parity'))"
ESM_AMD_OPERAND=f16x2 timeout 400 python bench.py --workload esm2_3b_contacts > $O/bench_3b_f16x2.log 2>&1; grep '^{' $O/bench_3b_f16x2.log > $O/bench_3b_f16x2.json; python -c "import json; r=json.load(open('$O/bench_3b_f16x2.json')); print('3b f16x2', r['value'], r['ms_per_step'], r.get('parity'))"
timeout 200 python tools/bench_vendor_gemm.py --smi > $O/vendor_gemm_calibration.log 2>&1; cat $O/vendor_gemm_calibration.log | tail -14
bash tools/profile_bench.sh r4final/prof_650m esm2_650m > $O/profile_650m.log 2>&1
bash tools/profile_bench.sh r4final/prof_msa msa1b > $O/profile_msa.log 2>&1
bash tools/profile_bench.sh r4final/prof_3b esm2_3b_contacts > $O/profile_3b.log 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/pmc_gemm9 -o g9 -- python tools/bench_gemm9.py --no-check --no-vendor --rounds 1 --iters 2 > $O/pmc_gemm9.log 2>&1
python tools/rocpd_pmc.py $(ls $O/pmc_gemm9/*/*_results.db $O/pmc_gemm9/*_results.db 2>/dev/null) > $O/pmc_gemm9_summary.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_gemm9f -o g9 -- python tools/bench_gemm9.py --no-check --no-vendor --rounds 1 --iters 2 > $O/pmc_gemm9f.log 2>&1
python tools/rocpd_pmc.py $(ls $O/pmc_gemm9f/*/*_results.db $O/pmc_gemm9f/*_results.db 2>/dev/null) > $O/pmc_gemm9_fetch_summary.txt 2>&1
find $O -name "*.db" -size +20M -delete
ls $O | head -50
