#!/usr/bin/env python3
"""Per-launch durations of the blur kernels of the last frame in the newest kernel trace under gpurun_out/kt."""
import csv
import glob
import os

fs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out/kt/**/*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"].split("(")[0].replace("void sifthip::", "").replace("sifthip::", ""),
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", "")) for r in rows]
idx = [i for i, n in enumerate(names) if n[0].startswith("first_kernel")]
tot = 0.0
for n in names[idx[-4]:]:
    if "blur" in n[0] or "first" in n[0] or "transpose" in n[0]:
        print("%-36s %8.1f us  grid %s x %s" % (n[0][:36], n[1], n[2], n[3]))
        tot += n[1]
print("sum %.1f us" % tot)
