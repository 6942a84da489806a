"""Fold the FETCH_SIZE pass and the WRITE_SIZE pass (rocprofv3 --pmc, rocpd sqlite output) into one JSON keyed by bench.py's kernel labels:
   hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches of the kernel.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide
coalesced reads (16 B/lane global loads and buffer_load ... lds alike: every read of these kernels) at 64 B, so the read side is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Infinity-Cache hits are counted, not excluded: this is fabric traffic behind L2, an
upper bound of the HBM bytes.
usage: python pmc_traffic_json.py <fetch.db> <write.db> <out.json>"""
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
from collections import defaultdict

TYPES = {"unsigned short": "bf16", "__half": "f16", "_Float16": "f16", "half": "f16", "float": "float"}      # as ops._timed names them


def demangle(names):
    """the rocpd database keeps mangled symbols (+ '.kd')"""
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    # (binutils' c++filt predates the _Float16 mangling "DF16_": spell it as the older half type "Dh")
    clean = [(n[:-3] if n.endswith(".kd") else n).replace("DF16_", "Dh") for n in names]
    out = subprocess.run([tool], input="\n".join(clean), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def label(name):
    """demangled kernel name -> the label ops._timed gives the launch (None: not a GEMM kernel of the table)"""
    m = re.match(r"(?:void )?(igemm_\w+)<([^>]*)>", name)
    if not m:
        return None
    kern, targs = m.group(1), [t.strip() for t in m.group(2).split(",")]
    t = TYPES.get(targs[0], targs[0])
    if kern == "igemm_nt8s_kernel":
        return f"igemm_nt8s_kernel<{t},{'patch' if targs[1] == 'true' else 'im2col'}>" + ("+splitk" if targs[2] == "true" else "")
    if kern == "igemm_nt_buf_kernel":
        return f"igemm_nt_buf_kernel<{t}>" + ("+splitk" if targs[2] == "true" else "")
    if kern == "igemm_nt_kernel":
        return f"igemm_nt_kernel<{t},{targs[1]},{targs[2]}>"
    if kern == "igemm_tn_kernel":
        return f"igemm_tn_kernel<{t}>"                      # all tile shapes of the generic TN kernel share one label
    return f"{kern}<{t}>"


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda k: [x for x in tabs if x.startswith("rocpd_" + k)][0]
    pmc = {r[0]: r[1] for r in cur.execute(f"select id, name from {t('info_pmc')}")}
    ksym = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {t('info_kernel_symbol')}")}
    plain = demangle(sorted(set(ksym.values())))
    ksym = {k: plain[v] for k, v in ksym.items()}
    disp = {r[0]: (r[1], r[2], r[3]) for r in cur.execute(f"select event_id, kernel_id, start, end from {t('kernel_dispatch')}")}
    tot, n, dur = defaultdict(float), defaultdict(int), defaultdict(float)
    per_event = defaultdict(float)
    for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from {t('pmc_event')}"):
        if ev in disp and pmc[pid] == counter:
            per_event[ev] += val                      # one row per XCD / instance: sum them
    for ev, val in per_event.items():
        kid, s, e = disp[ev]
        lab = label(ksym.get(kid, "?"))
        if lab is None:
            continue
        tot[lab] += val
        n[lab] += 1
        dur[lab] += e - s
    return {k: (tot[k] / n[k], n[k], dur[k] / n[k]) for k in tot}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch):
        f, nf, df = fetch[k]
        w, nw, dw = write.get(k, (0.0, 0, 0.0))
        out[k] = {"hbm_bytes_per_launch": round((2.0 * f + w) * 1024), "fetch_size_kib_raw": round(f, 1), "write_size_kib_raw": round(w, 1),
                  "launches_fetch_pass": nf, "launches_write_pass": nw, "avg_us_under_pmc": round(df / 1e3, 1),
                  "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024; gfx950 FETCH_SIZE counts wide reads at half their bytes"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:48s} {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch  ({v['launches_fetch_pass']} launches)")


if __name__ == "__main__":
    main()
