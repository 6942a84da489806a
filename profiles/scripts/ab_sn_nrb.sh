for rep in 1 2 3; do for v in 8 16; do
  cp ead-gan_amd/csrc/_ab/lib_nrb$v.so ead-gan_amd/libeadgan_hip.so
  out=$(timeout -k 10 120 python bench.py --no-probe --steps 80 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "SN_NRB=$v -> $out"
done; done
