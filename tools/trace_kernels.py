#!/usr/bin/env python3
"""Per-dispatch kernel durations of the last frame of a rocprofv3 --kernel-trace run:
    python tools/trace_kernels.py <dir> <frames> [substring]"""
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // int(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else ""
for r in rows[-n:]:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sifthip::", "")
    if sub in nm:
        print(f"{nm:40s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:8.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg {r['Workgroup_Size_X']}")
