O=gpurun_out/r3extra
mkdir -p $O
export TMPDIR=/tmp
for spec in "b4:--batch 4" "f16x2:--operand f16x2"; do
  tag=${spec%%:*}; a=${spec#*:}
  rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o bench -- python bench.py $a --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/trace_$tag.log 2>&1
  db=$(ls $O/trace_$tag/*/*_results.db $O/trace_$tag/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/kernel_stats_$tag.md 2>&1
done
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline > $O/repeat_$i.log 2>&1; python -c "import json; r=json.loads([l for l in open('$O/repeat_$i.log') if l.startswith('{')][-1]); print('repeat $i', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'])"; done
find $O -name "*.db" -delete
head -12 $O/kernel_stats_b4.md
