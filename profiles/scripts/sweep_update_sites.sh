for rep in 1 2 3; do for cfg in "g3 g3" "none g3" "g3 0" "none 0" "g3,d3 g3"; do set -- $cfg
  out=$(env EG_FUSE_ADAM_AT=$1 EG_BUCKET_OPT=$2 timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null)
  echo "fuse_adam_at=$1 bucket=$2 -> $out"
done; done
