#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes) of
tools/profile_extract.py into per-kernel HBM traffic per frame.

    python tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write <frames> > profiles/rNN_traffic.json

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE counts the 128-B requests of a streaming read as 64 B, so it is doubled;
WRITE_SIZE is taken as is.  Infinity-Cache hits are included in both.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def per_kernel(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg, n = collections.OrderedDict(), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sifthip::", "")
        if "kernel" not in k or "at::" in k or "rocclr" in k:
            continue
        agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
        n[k] += 1
    return agg, n


fetch, nf = per_kernel(sys.argv[1])
write, _ = per_kernel(sys.argv[2])
frames = int(sys.argv[3])
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sift_amd", "csrc", "pyramid.hip")
out = {"frames": frames, "unit": "bytes per frame", "correction": "2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024",
       # bench.py reports these figures only while the blur kernels are the ones that were measured
       "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(), "kernels": {}}
tot = 0.0
for k in fetch:
    rd = 2 * fetch[k] * 1024 / frames
    wr = write.get(k, 0.0) * 1024 / frames
    out["kernels"][k] = {"launches_per_frame": nf[k] / frames, "read": round(rd), "write": round(wr)}
    if k.startswith("blur_"):
        tot += rd + wr
out["pyramid_blur_total"] = round(tot)
print(json.dumps(out, indent=1))
