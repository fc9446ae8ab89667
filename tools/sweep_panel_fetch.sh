#!/bin/bash
# Fabric read traffic (FETCH_SIZE) and duration of the layer GEMMs against the tile-order panel width of the
# persistent kernel (ESMK_PANEL_C = N tiles per column panel; 0 = the launcher's choice, 5 for fc1's 20 N tiles).
# One rocprofv3 --pmc pass of a 2-step bench.py run per width (run on the GPU box through gpurun).
# usage: tools/sweep_panel_fetch.sh <outdir-under-gpurun_out> [widths...]
set -u
OUT=gpurun_out/${1:-panel}
shift
WIDTHS=${@:-0 1 2 10 20}
mkdir -p $OUT
export TMPDIR=/tmp
for P in $WIDTHS; do
  ESMK_PANEL_C=$P rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p$P -o bench -- \
      python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$P.log 2>&1
  db=$(ls $OUT/p$P/*/*_results.db $OUT/p$P/*_results.db 2>/dev/null | head -1)
  echo "== ESMK_PANEL_C=$P  (FETCH_SIZE in KiB of 32-byte-granular requests x 2 = bytes/1024, see tools/pmc_summary.py; _dur_us under the counter pass)"
  [ -n "$db" ] && python tools/rocpd_pmc.py $db 2>&1 | grep "gemm8_kernel\|attn_fwd" | head -8
  grep '^{' $OUT/p$P.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('   bench line under the profiler:', r['value'], 'residues/s', r['ms_per_step'], 'ms/step')" 2>/dev/null
  find $OUT/p$P -name "*.db" -delete
done
