"""Per-kernel MFMA utilisation out of two rocprofv3 --pmc passes of the same command (pmc_mfma.sh): pass 1 the SQ counters, pass 2
GRBM_GUI_ACTIVE.  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the chip's 1024 SIMDs (16 per v_mfma_f32_16x16x32_bf16);
GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so
    MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)
(1.0 = every SIMD's matrix pipe busy every cycle = the 2.5 PFLOP/s dense bf16 peak at the clock the launch ran at).  The SQ_WAIT_* /
SQ_ACTIVE_INST_ANY shares are fractions of SQ_WAVE_CYCLES (quad-cycles of resident waves): parked at s_waitcnt / barrier, stalled at issue,
issuing.  usage: python pmc_mfma.py <pass1.db> <pass2.db>"""
import shutil
import sqlite3
import subprocess
import sys
from collections import defaultdict


def demangle(names):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def load(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda k: [x for x in tabs if x.startswith("rocpd_" + k)][0]
    pmc_names = {r[0]: r[1] for r in cur.execute(f"select id, name from {t('info_pmc')}")}
    ksym = {r[0]: r[1].replace(".kd", "") for r in cur.execute(f"select id, kernel_name from {t('info_kernel_symbol')}")}
    dm = demangle(sorted(set(n for n in ksym.values() if "igemm" in n)))
    ksym = {k: dm.get(v, v) for k, v in ksym.items()}
    disp = {r[0]: (r[1], r[2], r[3], r[4], r[5]) for r in cur.execute(f"select event_id, kernel_id, start, end, grid_size_x, grid_size_z from {t('kernel_dispatch')}")}
    acc, cnt, dur, seen = defaultdict(lambda: defaultdict(float)), defaultdict(int), defaultdict(float), set()
    for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from {t('pmc_event')}"):
        if ev not in disp:
            continue
        kid, s, e, gx, gz = disp[ev]
        name = ksym.get(kid, "?").replace("void ", "").split("(")[0]
        if not name.startswith("igemm"):
            continue
        key = (name[:58], gx, gz)
        acc[key][pmc_names[pid]] += val
        if ev not in seen:
            seen.add(ev)
            cnt[key] += 1
            dur[key] += e - s
    return acc, cnt, dur


a1, c1, d1 = load(sys.argv[1])
a2, c2, d2 = load(sys.argv[2])
print(f"{'kernel':58s} {'grid':>14s} {'n':>3s} {'us':>7s} {'GHz':>5s} {'MFMA busy':>9s} {'wait':>6s} {'stall':>6s} {'issue':>6s} {'LDS confl':>9s}")
rows = []
for key in a1:
    if key not in a2:
        continue
    n1, n2 = c1[key], c2[key]
    g = a2[key]["GRBM_GUI_ACTIVE"] / n2 / 8.0
    v = {k: x / n1 for k, x in a1[key].items()}
    us = d1[key] / n1 / 1e3
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    rows.append((d1[key], f"{key[0]:58s} {str(key[1]) + 'x' + str(key[2]):>14s} {n1:3d} {us:7.1f} {g / (d2[key] / n2) :5.2f} {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * g):9.3f} "
                          f"{v.get('SQ_WAIT_ANY', 0) / wc:6.2f} {v.get('SQ_WAIT_INST_ANY', 0) / wc:6.2f} {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.2f} "
                          f"{v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 0), 1.0):9.3f}"))
for _, line in sorted(rows, reverse=True):
    print(line)
