"""Timeline analysis of a rocprofv3 --kernel-trace CSV of `bench.py`.  The trace is split into phases at idle gaps > GAP_US; for each phase:
span, union busy time, per-queue busy time, in-flight histogram, and the largest idle gaps with the kernels on either side.
usage: python profiles/scripts/timeline.py <kernel_trace.csv> [gap_us=300]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
GAP = (int(sys.argv[2]) if len(sys.argv) > 2 else 300) * 1000
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))


def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


phases, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] - end > GAP:
        phases.append(cur)
        cur = []
    cur.append(e)
    end = e[1] if end is None else max(end, e[1])
phases.append(cur)
print(f"{len(ev)} kernels, {len(phases)} phases (split at idle > {GAP / 1000:.0f} us)")
for pi, ph in enumerate(phases):
    s0, s1 = ph[0][0], max(e[1] for e in ph)
    span = s1 - s0
    if len(ph) < 50:
        print(f"phase {pi}: {len(ph)} kernels, span {span / 1e3:.0f} us  (skipped)")
        continue
    ub = union([(e[0], e[1]) for e in ph])
    adam = sum(1 for e in ph if e[2].startswith("adam_kernel"))
    print(f"phase {pi}: {len(ph)} kernels, span {span / 1e6:.2f} ms, union busy {ub / 1e6:.2f} ms = {100 * ub / span:.1f} %, sum of durations {sum(e[1] - e[0] for e in ph) / 1e6:.2f} ms, adam launches {adam}")
    byq = defaultdict(int)
    for e in ph:
        byq[(e[3], e[4])] += e[1] - e[0]
    print("   busy by (queue, stream):", {k: f"{100 * v / span:.1f}%" for k, v in sorted(byq.items(), key=lambda kv: -kv[1])})
    pts = []
    for e in ph:
        pts.append((e[0], 1)); pts.append((e[1], -1))
    pts.sort()
    k, last, hist = 0, pts[0][0], defaultdict(int)
    gaps = []
    for t, d in pts:
        hist[k] += t - last
        if k == 0 and t - last > 0:
            gaps.append((t - last, last, t))
        last = t
        k += d
    print("   in-flight -> share:", {kk: f"{100 * v / span:.1f}%" for kk, v in sorted(hist.items())})
    gs = sorted(gaps, reverse=True)
    print(f"   idle gaps: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.2f} ms; >20us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms; "
          f"5-20us: {sum(1 for g in gaps if 5000 < g[0] <= 20000)} totalling {sum(g[0] for g in gaps if 5000 < g[0] <= 20000) / 1e6:.2f} ms; <5us: {sum(g[0] for g in gaps if g[0] <= 5000) / 1e6:.2f} ms")
    for g in gs[:6]:
        before = max((e for e in ph if e[1] <= g[1] + 1), key=lambda e: e[1], default=None)
        after = min((e for e in ph if e[0] >= g[2] - 1), key=lambda e: e[0], default=None)
        print(f"      {g[0] / 1e3:7.1f} us  after {before[2][:48] if before else None!s:48}  before {after[2][:48] if after else None}")
    # gap attribution: idle time by the kernel that FOLLOWS the gap
    att = defaultdict(lambda: [0, 0])
    starts = sorted(ph, key=lambda e: e[0])
    import bisect
    st = [e[0] for e in starts]
    for g in gaps:
        i = bisect.bisect_left(st, g[2] - 1)
        if i < len(starts):
            a = att[starts[i][2][:56]]
            a[0] += g[0]; a[1] += 1
    print("   idle time by following kernel:")
    for n, (t, c) in sorted(att.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"      {t / 1e6:6.2f} ms in {c:4d} gaps (avg {t / c / 1e3:5.1f} us)  {n}")
