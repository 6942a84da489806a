#!/bin/bash
# generic A/B of one environment switch on the default bench, one box, alternating:  VAR=EG_X VARIANTS="a b" bash ab_env.sh [bench args]
for rep in 1 2 3; do for v in $VARIANTS; do
  out=$(env $VAR=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "$VAR=$v -> $out"
done; done
