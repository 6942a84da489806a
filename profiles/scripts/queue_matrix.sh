#!/bin/bash
# bench.py step time vs lane issue order (EG_DEFER) and the HIP runtime's graph / hardware queue counts.  Run from the repo root on the GPU box.
for defer in 0 1; do
  for gq in default 2 3 4 6 8; do
    for hq in default 8; do
      envs="EG_DEFER=$defer"
      [ "$gq" != default ] && envs="$envs DEBUG_HIP_FORCE_GRAPH_QUEUES=$gq"
      [ "$hq" != default ] && envs="$envs GPU_MAX_HW_QUEUES=$hq"
      out=$(env $envs timeout -k 10 120 python bench.py --no-probe --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
      echo "defer=$defer graph_queues=$gq hw_queues=$hq -> $out"
    done
  done
done
