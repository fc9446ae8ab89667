# PMC comparison of the plain residual kernel and the LayerNorm-fold producer (same K loop instruction stream, 2630 vs 3250
# cycles per K tile): tools/bench_gemm9.py --cases "fc2 resid" runs both; one counter group per pass
O=gpurun_out/r4k
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
CMD="python tools/bench_gemm9.py --no-vendor --no-check --cases fc2\ resid --rounds 1 --iters 2"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_CYCLES_VMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_BUSY_sum"; do
  i=$((i+1))
  eval rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o g9 -- $CMD > $O/p$i.log 2>&1
  python tools/rocpd_pmc.py $(ls $O/p$i/*/*_results.db $O/p$i/*_results.db 2>/dev/null) 2>&1 | grep "gemm9_kernel" > $O/p$i.txt
  echo "== group $i: $grp"; cat $O/p$i.txt | cut -c1-600; tail -2 $O/p$i.log | cut -c1-200
done
find $O -name "*.db" -delete
