#!/bin/bash
# the three secondary workloads: bench line + single-stream kernel statistics of each.  usage: bash profiles/scripts/small_nets.sh <tag>
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
run() { out=$(timeout -k 10 200 python bench.py --no-probe --no-cpu-baseline --no-roofline --steps 50 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null); echo "$* -> $out"; }
run --workload dsprites --dtype bf16 --batch 128
run --workload colored --dtype f16 --batch 512
run --workload mnist --dtype f32 --batch 256
cd /tmp && export TMPDIR=/tmp
for w in "dsprites bf16 128" "colored f16 512" "mnist f32 256"; do
  set -- $w
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$1 -o p -f csv -- python3 $R/bench.py --no-probe --workload $1 --dtype $2 --batch $3 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_${TAG}_$1.log 2>&1
  f=$(find $R/gpurun_out/prof_${TAG}_$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/${TAG}_$1_kernel_stats.csv
  t=$(find $R/gpurun_out/prof_${TAG}_$1 -name "*kernel_trace.csv" | head -1)
  gzip -f "$t"
done
