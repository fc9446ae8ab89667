#!/bin/bash
# PMC pass over the attention micro-benchmark for the kernel variants selected by ESMK_ATTN
# usage: tools/profile_attention.sh <outdir-under-gpurun_out> <variant> [<variant> ...]
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@"; do
  ESMK_ATTN=$v python tools/microbench.py --only attn --iters 20 > $OUT/time_$v.log 2>&1
  ESMK_ATTN=$v rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_$v -o a -- python tools/microbench.py --only attn --iters 3 > $OUT/pmc_$v.log 2>&1
  ESMK_ATTN=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $OUT/pmc2_$v -o a -- python tools/microbench.py --only attn --iters 3 > $OUT/pmc2_$v.log 2>&1
  echo "== ESMK_ATTN=$v"; cat $OUT/time_$v.log | tail -1
  python tools/rocpd_pmc.py $(ls $OUT/pmc_$v/*_results.db $OUT/pmc_$v/*/*_results.db $OUT/pmc2_$v/*_results.db $OUT/pmc2_$v/*/*_results.db 2>/dev/null) | grep attn
done
find $OUT -name "*.db" -size +5M -delete
