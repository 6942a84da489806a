"""Per-kernel PMC averages out of rocprofv3's sqlite output (rocpd schema): python pmc_table.py <results.db> [kernel-name-substring]"""
import sqlite3
import sys
from collections import defaultdict


def table(db, needle="igemm"):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda k: [x for x in tabs if x.startswith("rocpd_" + k)][0]
    pmc_names = {r[0]: r[1] for r in cur.execute(f"select id, name from {t('info_pmc')}")}
    ksym = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {t('info_kernel_symbol')}")}
    disp = {r[0]: (r[1], r[2], r[3], r[4]) for r in cur.execute(f"select event_id, kernel_id, start, end, grid_size_x from {t('kernel_dispatch')}")}
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    dur = defaultdict(float)
    seen = set()
    for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from {t('pmc_event')}"):
        if ev not in disp:
            continue
        kid, s, e, gx = disp[ev]
        name = ksym.get(kid, "?")
        if needle not in name:
            continue
        key = (name.split("(")[0][:70], gx)
        acc[key][pmc_names[pid]] += val
        if (ev,) not in seen:
            seen.add((ev,))
            cnt[key] += 1
            dur[key] += (e - s)
    for key in acc:
        n = cnt[key]
        print(f"{key[0]} grid {key[1]}  launches {n}  avg {dur[key] / n / 1e3:.1f} us")
        for k, v in sorted(acc[key].items()):
            print(f"    {k:32s} {v / n:16.0f}")


if __name__ == "__main__":
    table(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "igemm")
