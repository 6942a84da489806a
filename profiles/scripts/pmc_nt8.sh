cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/r02_counters.txt 2>&1
for cfg in 4:1 5:1 6:1 2:1; do
  tag=$(echo $cfg | tr ':' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/profiles/scripts/nt_layers.py --shapes AB --T 2 --configs $cfg --check 0 --rounds 1 --inner 3 > $R/gpurun_out/pmc_$tag.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --kernel-trace -d $R/gpurun_out/pmc2_$tag -o p -- python3 $R/profiles/scripts/nt_layers.py --shapes AB --T 2 --configs $cfg --check 0 --rounds 1 --inner 3 > $R/gpurun_out/pmc2_$tag.log 2>&1
done
ls -R $R/gpurun_out/pmc_4_1 | head -20
