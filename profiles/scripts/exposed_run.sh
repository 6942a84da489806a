# kernel trace of the default bench (hipGraph replay) -> which kernels run while no GEMM is in flight (exposed.py), per-queue phases (timeline.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_$TAG -o t -f csv -- python3 $R/bench.py --no-probe --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace_$TAG.log 2>&1 || { tail -3 $R/gpurun_out/trace_$TAG.log; exit 1; }
f=$(find $R/gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/scripts/exposed.py $f > $R/gpurun_out/${TAG}_exposed.txt 2>&1
cat $R/gpurun_out/${TAG}_exposed.txt
gzip -f $f
