#!/bin/bash
# rocprofv3 PMC passes over the contact-accumulation kernel on a 2-layer model of the 650M width (a full model
# under PMC serialisation takes minutes).  One counter group per pass; FETCH_SIZE / WRITE_SIZE need their own passes.
# usage: tools/profile_contacts.sh <outdir-under-gpurun_out> [batch]
set -u
OUT=gpurun_out/${1:-prof_ct}
B=${2:-64}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/bench_contacts.py --layers 2 --fused $B --materialised= --iters 1"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o ct -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_inst -o ct -- $CMD > $OUT/pmc_inst.log 2>&1
python tools/rocpd_pmc.py $(ls $OUT/pmc_*/*/*_results.db $OUT/pmc_*/*_results.db 2>/dev/null) > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*.db" -delete
grep -i "contact" $OUT/pmc_summary.txt
