#!/bin/bash
# optimizer updates bucket by bucket behind each bucket's own chain (EG_BUCKET_OPT=1, default) vs one whole-arena update behind all chains (0); one box, alternating
for rep in 1 2 3; do for v in ${VARIANTS:-0 1}; do
  out=$(env EG_BUCKET_OPT=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_BUCKET_OPT=$v -> $out"
done; done
