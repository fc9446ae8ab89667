# round 4, call 1: residual-epilogue desync A/B (microbench + forward), atomic ack probe
O=gpurun_out/r4a
mkdir -p $O
timeout 120 tools/build/atomic_probe > $O/atomic_probe.log 2>&1; cat $O/atomic_probe.log
timeout 400 python tools/bench_gemm9.py --no-vendor --no-check --cases resid --desync 0.15:0,0.3:0,0.5:0,0.75:0,0.5:1,0.5:2,1.0:2 --rounds 3 --iters 10 > $O/gemm9_desync.log 2>&1; cat $O/gemm9_desync.log
for d in 0 0.3 0.5 0.5:2; do
  f=${d%%:*}; g=0; [ "$d" != "$f" ] && g=${d#*:}
  ESMK_RESID_DESYNC=$f ESMK_RESID_DESYNC_GROUP=$g timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 12 --warmup 4 > $O/bench_desync_$d.log 2>&1
  python - <<PY
import json
for l in open("$O/bench_desync_$d.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print("desync $d:", r["value"], r["ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in r["kernel_classes"].items()})
PY
done
