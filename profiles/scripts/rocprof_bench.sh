# rocprofv3 kernel-trace summaries of the default bench command and of the single-stream one (--no-overlap); tag = first argument
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
for mode in default no_overlap; do
  extra=""; [ $mode = no_overlap ] && extra="--no-overlap"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$mode -o p -f csv -- python3 $R/bench.py --no-probe --steps 30 --warmup 5 --no-cpu-baseline --no-roofline $extra > $R/gpurun_out/prof_${TAG}_$mode.log 2>&1
  f=$(find $R/gpurun_out/prof_${TAG}_$mode -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/${TAG}_${mode}_kernel_stats.csv
  tail -1 $R/gpurun_out/prof_${TAG}_$mode.log | cut -c1-300
done
