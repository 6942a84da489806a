#!/bin/bash
# rocprofv3 kernel statistics of one secondary workload's default bench run.  usage: bash profiles/scripts/prof_small.sh <workload> <dtype> <batch> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$4 -o p -f csv -- python3 $R/bench.py --no-probe --workload $1 --dtype $2 --batch $3 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_$4.log 2>&1
f=$(find $R/gpurun_out/prof_$4 -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/$4_kernel_stats.csv
rm -rf $R/gpurun_out/prof_$4
