#!/usr/bin/env python3
"""Timeline of the last frame of a rocprofv3 --kernel-trace run (two streams): start, duration and
queue of every kernel, relative to the frame's first kernel.
    python tools/trace_timeline.py <dir> <frames>"""
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
rows = [r for r in rows if "sifthip" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // int(sys.argv[2])
fr = rows[-n:]
t0 = int(fr[0]["Start_Timestamp"])
qs = {}
for r in fr:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sifthip::", "")[:34]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{'    ' * q}q{q} {s / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  {nm} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
