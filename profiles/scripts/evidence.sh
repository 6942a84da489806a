#!/bin/bash
# one box, one call: the round's measurement set.  usage: bash profiles/scripts/evidence.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r02z}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default_run.json 2> gpurun_out/${TAG}_bench_default_run.err; echo "bench rc=$?"
cut -c1-300 gpurun_out/${TAG}_bench_default_run.json
bash profiles/scripts/table.sh > gpurun_out/${TAG}_throughput_table.txt 2>&1; cat gpurun_out/${TAG}_throughput_table.txt
bash profiles/scripts/rocprof_bench.sh $TAG
bash profiles/scripts/pmc_traffic.sh $TAG 2>&1 | tail -12
bash profiles/scripts/exposed_run.sh $TAG 2>&1 | tail -30
