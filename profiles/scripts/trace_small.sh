cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_ds -o t -f csv -- python3 $R/bench.py --no-probe --workload dsprites --dtype bf16 --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace_ds.log 2>&1
f=$(find $R/gpurun_out/trace_ds -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/scripts/trace_iter.py $f 15 > $R/gpurun_out/ds_iter.txt
rm -rf $R/gpurun_out/trace_ds
