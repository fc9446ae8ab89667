#!/bin/bash
# Round-6 evidence pass A (one box) for the library in the tree: smoke, rocprofv3 kernel stats + PMC passes (fold default at B = 64 and
# B = 4, plain at B = 64, 3B contacts, MSA-1b), then the driver's own line (its roofline.traffic finds these summaries).
# Output: gpurun_out/r6ev/
set -u
O=gpurun_out/r6ev
mkdir -p $O
T0=$(date +%s)
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_bench.sh r6ev/prof_650m esm2_650m > $O/profile_650m.log 2>&1; echo "prof 650m $(( $(date +%s) - T0 )) s"
bash tools/profile_bench.sh r6ev/prof_650m_b4 esm2_650m 4 1 > $O/profile_650m_b4.log 2>&1
bash tools/profile_bench.sh r6ev/prof_650m_plain esm2_650m 64 0 > $O/profile_650m_plain.log 2>&1; echo "prof plain $(( $(date +%s) - T0 )) s"
bash tools/profile_bench.sh r6ev/prof_650m_b4_plain esm2_650m 4 0 > $O/profile_650m_b4_plain.log 2>&1
bash tools/profile_bench.sh r6ev/prof_3b esm2_3b_contacts > $O/profile_3b.log 2>&1
bash tools/profile_bench.sh r6ev/prof_msa msa1b > $O/profile_msa.log 2>&1; echo "prof msa $(( $(date +%s) - T0 )) s"
# the summaries where bench.py looks for them (profiles/r6_pmc_summary_<workload>[_b<B>][_plain].json): the driver-like run below reports
# roofline.traffic from exactly these
cp $O/prof_650m/pmc_summary.json profiles/r6_pmc_summary_esm2_650m.json
cp $O/prof_650m_b4/pmc_summary.json profiles/r6_pmc_summary_esm2_650m_b4.json
cp $O/prof_650m_plain/pmc_summary.json profiles/r6_pmc_summary_esm2_650m_plain.json
cp $O/prof_650m_b4_plain/pmc_summary.json profiles/r6_pmc_summary_esm2_650m_b4_plain.json
cp $O/prof_3b/pmc_summary.json profiles/r6_pmc_summary_esm2_3b_contacts.json
cp $O/prof_msa/pmc_summary.json profiles/r6_pmc_summary_msa1b.json
mkdir -p $O/pmc_summaries; cp profiles/r6_pmc_summary_*.json $O/pmc_summaries/
T1=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_650m.log 2> $O/bench_650m.err; echo "default bench rc=$? wall $(( $(date +%s) - T1 )) s"
grep '^{' $O/bench_650m.log > $O/bench_650m.json
for spec in "b1:--batch 1 --steps 20 --warmup 5" "b8:--batch 8 --steps 20 --warmup 5" "b16:--batch 16" "b32:--batch 32" "bf16:--operand bf16"; do
  tag=${spec%%:*}; a=${spec#*:}
  timeout 300 python bench.py $a --no-secondary --no-cpu-baseline > $O/bench_650m_$tag.log 2>&1; grep '^{' $O/bench_650m_$tag.log > $O/bench_650m_$tag.json
  python -c "import json; r=json.load(open('$O/bench_650m_$tag.json')); print('$tag', r['value'], r['ms_per_step'], r['config'].get('ln_fold'))"
done
python - <<'PY'
import json
r = json.load(open('gpurun_out/r6ev/bench_650m.json'))
print('default', r['value'], r['ms_per_step'], r['config'].get('ln_fold'), r['roofline'])
for k, v in r.get('secondary_workloads', {}).items():
    print(' ', k, {x: v.get(x) for x in ('value', 'ms_per_step', 'wall_s', 'error', 'skipped')}, (v.get('roofline') or {}).get('traffic'))
PY
echo "total $(( $(date +%s) - T0 )) s"
