#!/bin/bash
# 64 rows per pipeline step for the single-output-tile weight-gradient launches (EG_TN_KR64=1) vs 32 (0)
for w in "dsprites bf16 128" "mnist bf16 128" "colored bf16 512"; do set -- $w; for v in 0 1 0 1; do
  out=$(env EG_TN_KR64=$v timeout -k 10 120 python bench.py --no-probe --workload $1 --dtype $2 --batch $3 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "$1 EG_TN_KR64=$v -> $out"
done; done
