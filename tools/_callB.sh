O=gpurun_out/r3B
mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $O/pol_$tag.log 2>&1; python -c "import json; r=json.loads([l for l in open('$O/pol_$tag.log') if l.startswith('{')][-1]); print('$tag', r['value'], r['ms_per_step'], {k: v['ms_per_step'] for k, v in r['kernel_classes'].items() if v['ms_per_step'] > 0.5})"; }
for b in 1 2 4 8 16 32; do
  ESMK_GEMM9_POLICY=0 run b${b}_pol0 --batch $b
  run b${b}_pol1 --batch $b
done
