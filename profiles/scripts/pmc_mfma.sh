# MFMA-busy fraction, wait / issue-stall shares and LDS bank conflicts of the GEMM kernels that ship, measured IN the CelebA step (eager,
# single stream: `bench.py --no-graph --no-overlap`): two rocprofv3 --pmc passes (SQ block; GRBM block), the program directly after `--`.
# usage: bash profiles/scripts/pmc_mfma.sh <tag>   -> gpurun_out/<tag>_pmc_mfma.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 420 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcm_${TAG}_$i -o p -- python3 $R/bench.py --no-probe --steps 3 --warmup 2 --no-graph --no-overlap --no-cpu-baseline --no-roofline > $R/gpurun_out/pmcm_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmcm_${TAG}_$i.log; exit 1; }
  echo "pass $i done"
done
python3 $R/profiles/scripts/pmc_mfma.py $(find $R/gpurun_out/pmcm_${TAG}_1 -name "*.db" | head -1) $(find $R/gpurun_out/pmcm_${TAG}_2 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_pmc_mfma.txt
cat $R/gpurun_out/${TAG}_pmc_mfma.txt
