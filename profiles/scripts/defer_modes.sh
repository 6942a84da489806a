#!/bin/bash
for defer in 0 once 1; do
  out=$(env EG_DEFER=$defer timeout -k 10 120 python bench.py --no-probe --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_DEFER=$defer -> $out"
done
