# short form of tools/final_evidence.sh for a library change late in a round: tests, smoke, the driver's bench line, the small-batch lines, rocprofv3 stats + PMC passes of the three workloads.  outputs under gpurun_out/r4final2
O=gpurun_out/r4final2
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
grep -E "rel|err|L2|argmax|logit|floor|MSA \(|MSA 4|dims|worst|consumer|folded|fold " $O/pytest_gpu.log | grep -v "^tests/" > $O/gpu_tests_parity_lines.txt
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s); timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_650m.log 2>&1; echo "default bench rc=$? wall $(( $(date +%s) - T0 )) s"
grep '^{' $O/bench_650m.log > $O/bench_650m.json
bash tools/profile_bench.sh r4final2/prof_650m esm2_650m > $O/profile_650m.log 2>&1
for spec in "b1:--batch 1" "b4:--batch 4" "b4_lnfold:--batch 4 --ln-fold 1" "b8:--batch 8"; do
  tag=${spec%%:*}; a=${spec#*:}
  timeout 200 python bench.py $a --no-secondary --no-cpu-baseline > $O/bench_650m_$tag.log 2>&1; grep '^{' $O/bench_650m_$tag.log > $O/bench_650m_$tag.json
  python -c "import json; r=json.load(open('$O/bench_650m_$tag.json')); print('$tag', r['value'], r['ms_per_step'])"
done
bash tools/profile_bench.sh r4final2/prof_msa msa1b > $O/profile_msa.log 2>&1
bash tools/profile_bench.sh r4final2/prof_3b esm2_3b_contacts > $O/profile_3b.log 2>&1
ls $O/prof_*/pmc_summary.json
