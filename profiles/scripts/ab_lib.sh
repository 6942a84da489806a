#!/bin/bash
# A/B of two library builds (ead-gan_amd/csrc/_ab/lib_a.so = before, lib_b.so = after) on one box, alternating:  bash ab_lib.sh [bench args]
for rep in 1 2 3; do for v in a b; do
  cp ead-gan_amd/csrc/_ab/lib_$v.so ead-gan_amd/libeadgan_hip.so
  out=$(timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "lib_$v -> $out"
done; done
