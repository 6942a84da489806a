#!/bin/bash
# NT planner thresholds under the final schedule: --igemm-tuning buf_min_tiles,splitk_target,big_min_tiles,persistent,wide_min_tiles
for t in "512,512,0,0,0" "256,512,0,0,0" "128,512,0,0,0" "64,512,0,0,0" "1,512,0,0,0" "256,512,0,0,0" "512,512,0,0,0"; do
  out=$(timeout -k 10 120 python bench.py --no-probe --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --igemm-tuning $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "--igemm-tuning $t -> $out"
done
