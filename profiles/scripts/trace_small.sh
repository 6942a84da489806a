#!/bin/bash
# one iteration of a secondary workload as a timeline (kernel trace of the default bench command, profiles/scripts/trace_iter.py)
# usage: bash profiles/scripts/trace_small.sh <workload> <dtype> <batch>     -> gpurun_out/<workload>_iter.txt, gpurun_out/<workload>_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-dsprites}; D=${2:-bf16}; B=${3:-128}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_$W -o t -f csv -- python3 $R/bench.py --no-probe --workload $W --dtype $D --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace_$W.log 2>&1
f=$(find $R/gpurun_out/trace_$W -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/scripts/trace_iter.py $f 15 > $R/gpurun_out/${W}_iter.txt
cp $(find $R/gpurun_out/trace_$W -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${W}_stats.csv
rm -rf $R/gpurun_out/trace_$W
