"""Per (kernel, grid) durations of the 8-wave NT kernels in a rocprofv3 --kernel-trace CSV of `bench.py --no-overlap` (one stream): the per-shape
view that does not depend on host launch latency (bench.py's EG_BENCH_DETAIL table brackets eager launches with HIP events: a launch that
follows short kernels finds the GPU idle and its bracket then includes the host's launch time).
usage: python profiles/scripts/gemm_by_grid.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "igemm_nt8s" in n or "igemm_nt8h" in n or "igemm_tn8" in n:
        base = "igemm_nt8s" if "nt8s" in n else ("igemm_nt8h" if "nt8h" in n else "igemm_tn8")
        stat = "+stat" if base != "igemm_tn8" and n.split(">")[0].rstrip().endswith("true") else ""
        d[(base + stat, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':18s} {'grid (threads x, y, z)':>24s} {'n':>4s} {'median us':>10s} {'min':>7s} {'max':>7s}")
for k, v in sorted(d.items()):
    v = sorted(v)
    print(f"{k[0]:18s} {str(k[1:]):>24s} {len(v):4d} {v[len(v) // 2]:10.1f} {v[0]:7.1f} {v[-1]:7.1f}")
