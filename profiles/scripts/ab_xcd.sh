#!/bin/bash
# XCD-aware M-tile order in igemm_nt_buf (EG_XCD_REMAP=1: each XCD gets a contiguous range of M tiles) vs dispatch order (0); one box, alternating
for rep in 1 2 3; do for v in 0 1; do
  out=$(env EG_XCD_REMAP=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" 2>/dev/null)
  echo "EG_XCD_REMAP=$v -> $out"
done; done
