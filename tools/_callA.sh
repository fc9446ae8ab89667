# experiments: K-stagger / DMA-bound hypotheses in gemm9, MFMA shape energy probe, B=4 fork, B=4 K-step cycles
O=gpurun_out/r3A
mkdir -p $O
timeout 120 hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/probe > $O/probe_build.log 2>&1
timeout 120 /tmp/probe > $O/mfma_power_probe.log 2>&1; cat $O/mfma_power_probe.log
timeout 400 python tools/bench_gemm9.py --no-check --no-vendor --rounds 3 --iters 8 --dbg --cases "qk store,fc2 store,v/out store" > $O/gemm9_dma_variants.log 2>&1; grep -v "^device" $O/gemm9_dma_variants.log
timeout 200 python tools/bench_gemm9.py --B 4 --no-check --no-vendor --rounds 3 --iters 20 > $O/gemm9_b4.log 2>&1; cat $O/gemm9_b4.log
for f in 0 1; do
  for b in 4 16; do
    ESMK_QKV_FORK=$f timeout 200 python bench.py --batch $b --no-secondary --no-cpu-baseline > $O/bench_b${b}_fork$f.log 2>&1
    python -c "import json; r=json.loads([l for l in open('$O/bench_b${b}_fork$f.log') if l.startswith('{')][-1]); print('B=$b fork=$f', r['value'], r['ms_per_step'], {k: v['ms_per_step'] for k, v in r['kernel_classes'].items()})"
  done
done
