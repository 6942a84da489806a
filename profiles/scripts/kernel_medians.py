"""median duration per (kernel, grid) of a rocprofv3 --kernel-trace CSV.  usage: python profiles/scripts/kernel_medians.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    print(f"{k[0]:72s} g{k[1]}x{k[2]} n={len(v)} median {v[len(v) // 2]:.1f} us")
