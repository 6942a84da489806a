# HBM traffic per launch of the step's GEMM kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one
# pass, MI355X_MICROARCH.md counters table), eager single-stream launches of the default bench workload; then
# profiles/scripts/pmc_traffic_json.py folds both databases into profiles/<tag>_pmc_traffic.json.   usage: bash pmc_traffic.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
shift
# further arguments go to bench.py (e.g. --workload mnist --dtype f32 --batch 256); output name: <tag>_pmc_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 420 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmct_${TAG}_$c -o p -- python3 $R/bench.py --no-probe --steps 3 --warmup 2 --no-graph --no-overlap --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/pmct_${TAG}_$c.log 2>&1 || { echo "pass $c failed"; tail -3 $R/gpurun_out/pmct_${TAG}_$c.log; exit 1; }
  echo "pass $c done"
done
python3 $R/profiles/scripts/pmc_traffic_json.py $(find $R/gpurun_out/pmct_${TAG}_FETCH_SIZE -name "*.db" | head -1) $(find $R/gpurun_out/pmct_${TAG}_WRITE_SIZE -name "*.db" | head -1) $R/gpurun_out/${TAG}_pmc_traffic.json
