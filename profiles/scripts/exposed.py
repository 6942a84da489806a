"""Which kernels run with NO GEMM in flight during the hipGraph replays of `bench.py`?  Input: rocprofv3 --kernel-trace CSV (optionally .gz).
usage: python profiles/scripts/exposed.py <kernel_trace.csv[.gz]>"""
import collections
import csv
import gzip
import sys

f = sys.argv[1]
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Stream_Id"]) for r in rows))
g = [e for e in ev if e[4] == "0" and e[3] not in ("1",)]          # graph branches land on extra queues of stream 0
t0, t1 = g[0][0], g[-1][1]
ph = [e for e in ev if e[0] >= t0 and e[1] <= t1]
once = sum(1 for e in ph if "info_losses_rpqxy" in e[2]) or sum(1 for e in ph if "affine_reg_rpqxy" in e[2])   # one launch per CelebA iteration (the info step's loss kernel)
steps = once if once else sum(1 for e in ph if "adam_tick" in e[2]) / 3
print(f"launches per step: {len(ph) / steps:.1f}")
span = t1 - t0
print(f"replay window {span / 1e6:.2f} ms, {steps:.1f} steps, {span / 1e6 / steps:.3f} ms/step")
isg = lambda n: "igemm" in n
pts = []
for e in ph:
    pts.append((e[0], 1, isg(e[2]), e[2])); pts.append((e[1], -1, isg(e[2]), e[2]))
pts.sort(key=lambda p: (p[0], p[1]))
ng = ns = 0
last = pts[0][0]
acc, alone, cur = collections.Counter(), collections.Counter(), collections.Counter()
for t, d, gm, name in pts:
    dt = t - last
    acc["idle" if ng + ns == 0 else ("gemm only" if ns == 0 else ("small only" if ng == 0 else "gemm+small"))] += dt
    if ng == 0 and ns > 0:
        for n, c in cur.items():
            if c > 0:
                alone[n[:60]] += dt
    last = t
    if gm:
        ng += d
    else:
        ns += d
        cur[name] += d
for k, v in acc.items():
    print(f"{k:12s} {v / 1e6 / steps:6.3f} ms/step  {100 * v / span:5.1f}%")
print("time with only small kernels in flight, by kernel (ms/step):")
for k, v in alone.most_common(22):
    print(f"  {v / 1e6 / steps:6.3f}  {k}")
