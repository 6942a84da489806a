#!/bin/bash
# the round's whole measurement set on one box.  PMC traffic passes first (bench.py's roofline object reads the committed
# profiles/<round>_pmc_traffic*.json of this code), then the CelebA headline (default bench line, throughput table, rocprofv3 kernel
# statistics of the default and the single-stream run, exposed-time analysis, per-shape step detail), then the three secondary workloads.
# usage: bash profiles/scripts/evidence_all.sh <tag>     (tag = round, e.g. r03: files gpurun_out/<tag>_* and profiles/<tag>_pmc_traffic*.json)
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/scripts/pmc_traffic.sh $TAG 2>&1 | tail -8
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
for w in "mnist f32 256" "dsprites bf16 128" "colored f16 512"; do
  set -- $w
  bash profiles/scripts/pmc_traffic.sh ${TAG}_$1 --workload $1 --dtype $2 --batch $3 2>&1 | tail -2
  cp gpurun_out/${TAG}_$1_pmc_traffic.json profiles/${TAG}_pmc_traffic_$1.json
done
cd $R
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default_run.json 2> gpurun_out/${TAG}_bench_default_run.err; echo "bench rc=$?"
cut -c1-300 gpurun_out/${TAG}_bench_default_run.json
bash profiles/scripts/table.sh > gpurun_out/${TAG}_throughput_table.txt 2>&1; cat gpurun_out/${TAG}_throughput_table.txt
bash profiles/scripts/rocprof_bench.sh $TAG
bash profiles/scripts/exposed_run.sh $TAG 2>&1 | tail -30
cd $R
EG_BENCH_DETAIL=1 timeout -k 10 300 python bench.py --no-probe --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2> gpurun_out/${TAG}_step_detail.txt; echo "detail rc=$?"
for w in "mnist f32 256" "dsprites bf16 128" "colored f16 512"; do
  set -- $w
  timeout -k 10 300 python bench.py --no-probe --workload $1 --dtype $2 --batch $3 > gpurun_out/${TAG}_bench_$1.json 2> gpurun_out/${TAG}_bench_$1.err; echo "$1 rc=$?"
  cut -c1-200 gpurun_out/${TAG}_bench_$1.json
done
