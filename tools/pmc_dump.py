#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 --pmc output directory (kernels whose name contains the filter):
    python tools/pmc_dump.py <dir> [filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void sifthip::", "")
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
for k, v in rows.items():
    print(k)
    for a, b in sorted(v.items()):
        print("   %-28s %14.0f  (%d dispatches)" % (a, b / n[(k, a)], n[(k, a)]))
