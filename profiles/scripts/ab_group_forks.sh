#!/bin/bash
# fork grouping A/B (EG_GROUP_FORKS=1: chains of two adjacent layers share one fork).  The switch was removed from the code after this
# measurement (slower: profiles/r01_timeline_notes.md item 7); kept as the record of how it was measured.
for rep in 1 2 3; do for v in 0 1; do
  out=$(env EG_GROUP_FORKS=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_GROUP_FORKS=$v -> $out"
done; done
