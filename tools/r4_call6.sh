O=gpurun_out/r4f
mkdir -p $O
for h in 0 1 2 3 4 8 12; do
  ESMK_ATTN_HACK=$h timeout 200 python bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_hack$h.log 2>&1
  grep '^{' $O/bench_hack$h.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('attention hack $h:', r['kernel_classes']['attention']['ms_per_step'], 'ms/step; step', r['ms_per_step'])"
done
