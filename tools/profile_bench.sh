#!/bin/bash
# rocprofv3 passes over a short bench.py run (run on the GPU box through gpurun):
#   1. kernel trace + stats            -> per-kernel durations
#   2-4. PMC passes (kernel trace only, one counter group per pass, as the microarch guide prescribes)
# usage: tools/profile_bench.sh <outdir-under-gpurun_out> [workload] [batch] [ln_fold 0|1]
#   workload: esm2_650m (default) | msa1b | esm2_3b_contacts;  batch: per-GPU batch (default: the workload's);
#   ln_fold: 1 (default: the library default, fold on) | 0 (--ln-fold 0)
# The summary (pmc_summary.json) records workload, batch, fold mode and the library's source hash: bench.py reports
# roofline.traffic only for exactly that combination.
set -u
OUT=gpurun_out/${1:-prof}
WL=${2:-esm2_650m}
case $WL in esm2_650m) DB=64 ;; esm2_3b_contacts) DB=32 ;; msa1b) DB=1 ;; *) DB=64 ;; esac
B=${3:-$DB}
FOLD=${4:-1}
mkdir -p $OUT
export TMPDIR=/tmp
# per-launch figures describe the launches of the ONE-stream forward (bench.py's per-class HIP-event profile also runs one stream:
# esm_amd/esm2.py profile_begin); the dual-stream split of small batches (ESM_AMD_DUAL_STREAM) only changes the timed `value`
export ESM_AMD_DUAL_STREAM=0
CMD="python bench.py --workload $WL --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
[ "$FOLD" = "0" ] && CMD="$CMD --ln-fold 0"
[ "$WL" = "msa1b" ] && FOLDARG="" || FOLDARG="--ln-fold $FOLD"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o bench -- $CMD > $OUT/pmc_sq.log 2>&1
for d in trace; do
  db=$(ls $OUT/$d/*/*_results.db $OUT/$d/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $OUT/kernel_stats.md 2>&1
done
python tools/rocpd_pmc.py $(ls $OUT/pmc_*/*/*_results.db $OUT/pmc_*/*_results.db 2>/dev/null) > $OUT/pmc_summary.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_summary.json --workload $WL --batch $B $FOLDARG $(ls $OUT/pmc_*/*/*_results.db $OUT/pmc_*/*_results.db 2>/dev/null) > $OUT/pmc_summary_json.log 2>&1
ls -R $OUT | head -40 > $OUT/files.txt
# the raw traces are large: keep only the summaries
find $OUT -name "*.db" -size +20M -delete
tail -3 $OUT/*.log
cat $OUT/kernel_stats.md | head -30
cat $OUT/pmc_summary.txt | head -30
