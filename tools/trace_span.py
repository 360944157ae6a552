#!/usr/bin/env python3
"""Per-frame timeline of a rocprofv3 --kernel-trace run of tools/profile_extract.py:
span from the first to the last kernel of each frame, busy time per queue, idle gaps.
    python tools/trace_span.py <dir> <frames>"""
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
rows = [r for r in rows if "sifthip" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames = int(sys.argv[2])
n = len(rows) // frames
for f in range(frames):
    fr = rows[f * n:(f + 1) * n]
    t0 = int(fr[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in fr)
    # union of busy intervals
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in fr)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per_q = {}
    for r in fr:
        per_q[r["Queue_Id"]] = per_q.get(r["Queue_Id"], 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    nxt = int(rows[(f + 1) * n]["Start_Timestamp"]) if f + 1 < frames else None
    print(f"frame {f}: span {(t1 - t0) / 1e3:.0f} us, device busy (union) {busy / 1e3:.0f} us, "
          f"per queue {[round(v / 1e3) for v in per_q.values()]} us"
          + (f", gap to next frame {(nxt - t1) / 1e3:.0f} us" if nxt else ""))
