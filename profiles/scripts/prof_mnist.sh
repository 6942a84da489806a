cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_zu_mnist -o p -f csv -- python3 $R/bench.py --no-probe --workload mnist --dtype f32 --batch 256 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_zu_mnist.log 2>&1
f=$(find $R/gpurun_out/prof_zu_mnist -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r03zu_mnist_kernel_stats.csv
rm -rf $R/gpurun_out/prof_zu_mnist
