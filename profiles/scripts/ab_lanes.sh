#!/bin/bash
# number of weight-gradient lanes (EG_LANES): alternating runs on one box
for rep in 1 2; do for v in 2 3 4 6; do
  out=$(env EG_LANES=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_LANES=$v -> $out"
done; done
