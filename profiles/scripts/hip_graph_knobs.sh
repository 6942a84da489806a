#!/bin/bash
# HIP runtime graph knobs found in libamdhip64 (undocumented): effect on the replayed CelebA step, one box
run() { out=$(env "$@" timeout -k 10 120 python bench.py --no-probe --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['workload'][-44:-20])" 2>/dev/null); echo "$* -> $out"; }
run A=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=16
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run A=1
