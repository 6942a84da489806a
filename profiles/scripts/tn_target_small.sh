#!/bin/bash
for w in "colored f16 512" "dsprites bf16 128"; do set -- $w; for t in 768 256 384 512 768 1024; do
  out=$(env EG_TN_TARGET=$t timeout -k 10 120 python bench.py --no-probe --workload $1 --dtype $2 --batch $3 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "$1 EG_TN_TARGET=$t -> $out"
done; done
