#!/bin/bash
# multi-graph replay x number of HSA queues per process
for rep in 1 2; do for mg in 0 1; do for hq in 4 6 8 12; do
  out=$(env EG_MULTI_GRAPH=$mg GPU_MAX_HW_QUEUES=$hq timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "multi_graph=$mg hw_queues=$hq -> $out"
done; done; done
