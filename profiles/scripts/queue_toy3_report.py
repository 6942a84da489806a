"""timeline of the LAST graph replay in a queue_toy3 trace: main kernels have grid < 2 * lane kernels' grid"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), r["Queue_Id"]) for r in rows if "add" in r["Kernel_Name"] or "elementwise" in r["Kernel_Name"])
ev = ev[-16:]
t0 = ev[0][0]
for s, e, gsz, q in ev:
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{q} grid {gsz} {'LANE' if gsz > 24 * 1024 * 1024 // 4 * 0 + 6000000 else 'main'}")
