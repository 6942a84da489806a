#!/bin/bash
# image-side transposed convolutions: one GEMM + col2im (EG_IMG_GEMM=1) vs the 4-phase implicit GEMM (0); alternating runs on one box
for rep in 1 2 3; do for v in 0 1; do
  out=$(env EG_IMG_GEMM=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_IMG_GEMM=$v -> $out"
done; done
