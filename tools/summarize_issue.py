#!/usr/bin/env python3
"""Issue-slot utilisation of the blur kernels of one level from a rocprofv3 counter pass with a kernel trace:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d <dir> -- python3 tools/profile_blur_level.py
    python tools/summarize_issue.py <dir> > profiles/rNN_issue.json
issue utilisation = SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs); the file carries the hash of
pyramid.hip so that bench.py quotes it only for the sources it was measured on."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "blur" not in r["Kernel_Name"]:
            continue
        key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void sifthip::", ""))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
acc = collections.OrderedDict()
for (did, name), c in rows.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    a = acc.setdefault(name, {"launches": 0, "us": 0.0, "valu": 0.0, "util": 0.0, "waves": 0.0, "ghz": 0.0})
    a["launches"] += 1
    a["us"] += dur.get(did, 0.0)
    a["valu"] += c["SQ_INSTS_VALU"]
    a["util"] += c["SQ_INSTS_VALU"] * 4 / 1024 / cyc
    a["waves"] += c["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc
    a["ghz"] += cyc / dur.get(did, 1.0) / 1e3
out = {"workload": "one exact-mode blur level, sigma 6.4 (39 taps), 7680x4320 (octave 0 of a 4K frame)",
       "kernel_source_sha256": hashlib.sha256(open(os.path.join(root, "sift_amd", "csrc", "pyramid.hip"), "rb").read()).hexdigest(),
       "definition": "issue_utilisation = SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs)",
       "kernels": {}}
for name, a in acc.items():
    n = a["launches"]
    out["kernels"][name] = {"launches": n, "us": round(a["us"] / n, 1), "valu_instructions": round(a["valu"] / n),
                            "issue_utilisation": round(a["util"] / n, 3), "waves_per_simd_mean": round(a["waves"] / n, 2),
                            "clock_ghz": round(a["ghz"] / n, 2)}
print(json.dumps(out, indent=1))
