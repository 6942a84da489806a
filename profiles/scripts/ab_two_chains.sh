#!/bin/bash
# the discriminator step as a second chain beside the rest of step 1 (EG_TWO_CHAINS=1) vs after it on the main stream (0); one box, alternating.
# The switch was removed from the code after this measurement (slower: profiles/r01_timeline_notes.md item 10); kept as the record.
for rep in 1 2 3; do for v in 0 1; do
  out=$(env EG_TWO_CHAINS=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_TWO_CHAINS=$v -> $out"
done; done
