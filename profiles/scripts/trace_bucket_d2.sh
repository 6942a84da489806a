cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export EG_FUSE_ADAM_AT=d2,g3 EG_BUCKET_OPT=d2,g3
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_r03w -o t -f csv -- python3 $R/bench.py --no-probe --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace_r03w.log 2>&1
f=$(find $R/gpurun_out/trace_r03w -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/scripts/trace_iter.py $f 15 > $R/gpurun_out/r03w_iter_bucket_d2.txt
gzip -f $f
