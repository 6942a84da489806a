# PMC passes over the NT variants on one balanced launch shape (B T=2: 256 workgroups of 256x128, 64 K tiles); one counter block per pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${SHAPE:-B}; T=${T:-2}
for cfg in ${CFGS:-7:1 8:1 2:1}; do
  tag=$(echo $cfg | tr ':' '_')
  i=0
  for set in "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" \
             "TA_BUSY TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
             "TD_TD_BUSY TD_TC_STALL TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" \
             "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES TCP_TOTAL_READ" \
             "TA_BUFFER_READ_LDS_WAVEFRONTS TA_BUFFER_TOTAL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcs_${tag}_$i -o p -- python3 $R/profiles/scripts/nt_layers.py --shapes $SHAPE --T $T --configs $cfg --check 0 --rounds 1 --inner 3 > $R/gpurun_out/pmcs_${tag}_$i.log 2>&1
    python3 $R/profiles/scripts/pmc_table.py $R/gpurun_out/pmcs_${tag}_$i/p_results.db igemm_nt
  done
done
