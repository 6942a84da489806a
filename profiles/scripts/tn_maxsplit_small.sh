#!/bin/bash
# split cap of the per-tap weight-gradient kernel on the small networks at HEAD (the image-side layers have one output tile: the cap is their grid)
for w in "colored f16 512" "dsprites bf16 128" "mnist f32 256"; do set -- $w; for v in 128 256 512 1024 128 512; do
  out=$(env EG_TN_MAXSPLIT=$v timeout -k 10 120 python bench.py --no-probe --workload $1 --dtype $2 --batch $3 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "$1 EG_TN_MAXSPLIT=$v -> $out"
done; done
