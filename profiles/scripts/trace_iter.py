"""One iteration of a rocprofv3 --kernel-trace CSV (optionally .gz) of bench.py as a timeline: start (us from the iteration's first kernel),
duration, queue, grid, kernel name.  usage: python profiles/scripts/trace_iter.py <kernel_trace.csv[.gz]> [iteration=15]"""
import csv
import gzip
import sys

f = sys.argv[1]
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Grid_Size_X"], r["Grid_Size_Z"]) for r in rows)
idx = [i for i, e in enumerate(ev) if e[2].startswith("rng_fill")]
starts = [idx[i] for i in range(len(idx)) if i == 0 or idx[i] - idx[i - 1] > 20]
it = int(sys.argv[2]) if len(sys.argv) > 2 else 15
s, e = starts[it], starts[it + 1]
t0 = ev[s][0]
print(f"iteration {it}: {(ev[e][0] - t0) / 1e3:.1f} us, {e - s} launches")
for k in ev[s:e]:
    name = k[2].replace("void ", "").split("(")[0][:60]
    print(f"{(k[0] - t0) / 1e3:8.1f} {(k[1] - k[0]) / 1e3:7.1f}  q{k[3]} g{k[4]}x{k[5]} {name}")
