#!/bin/bash
# re-packing after an optimizer update split over two pack lanes (EG_PACK_LANES=1) vs one chain on the optimizer lane (0); one box, alternating.
# The switch was removed from the code after this measurement (slower: profiles/r01_timeline_notes.md item 11); kept as the record.
for rep in 1 2 3; do for v in 0 1; do
  out=$(env EG_PACK_LANES=$v timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "EG_PACK_LANES=$v -> $out"
done; done
