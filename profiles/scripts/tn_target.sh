#!/bin/bash
# whole-step time vs the weight-gradient (TN) kernel's workgroup target (split-K along the batch rows)
for t in 384 512 640 768 1024 1536; do
  out=$(env EG_TN_TARGET=$t timeout -k 10 120 python bench.py --no-probe --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_TN_TARGET=$t -> $out"
done
