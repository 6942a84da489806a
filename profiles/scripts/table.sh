#!/bin/bash
# the DESIGN.md throughput table: one bench.py line per row (no cpu baseline / roofline pass)
run() { out=$(env ${EXTRA:-A=1} timeout -k 10 200 python bench.py --no-probe --no-cpu-baseline --no-roofline --steps 50 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null); echo "$* -> $out"; }
run --workload celeba --dtype bf16 --batch 128
run --workload celeba --dtype bf16 --batch 128 --no-overlap
run --workload celeba --dtype f16 --batch 128
run --workload celeba --dtype bf16 --batch 512
run --workload celeba --dtype bf16 --batch 128 --resident-inputs
run --workload celeba --dtype bf16 --batch 128 --no-graph
run --workload celeba --dtype bf16 --batch 128 --force-dist
EXTRA="MASTER_PORT=29531" run --workload celeba --dtype bf16 --batch 128 --force-dist --wire bf16     # (own port: the previous row's socket may still be closing)
run --workload mnist --dtype f32 --batch 256
run --workload mnist --dtype bf16 --batch 128
run --workload dsprites --dtype bf16 --batch 128
run --workload dsprites --dtype f32 --batch 128
run --workload colored --dtype f16 --batch 512
run --workload colored --dtype bf16 --batch 512
run --workload pxy --dtype bf16 --batch 128
run --workload pxy --dtype f32 --batch 128
