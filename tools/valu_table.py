#!/usr/bin/env python3
"""Wave-level instruction counts per kernel from one rocprofv3 counter pass:
    SIFT_HIP_SERIAL=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES \\
        --output-format csv -d gpurun_out/pmc_valu -- python3 tools/profile_extract.py --steps 2
    python tools/valu_table.py gpurun_out/pmc_valu 2 > profiles/rNN_valu_instructions.txt"""
import collections
import csv
import glob
import sys

d, frames = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void sifthip::", "").replace("sifthip::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            launches[k] += 1
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values()) / frames
print("wave-level instructions per kernel, per frame (%d profiled frames)\n" % frames)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    print("%-34s launches/frame %3d  VALU %7.2f M (%4.1f%%)  SALU %6.2f M  LDS %6.2f M  VMEM_RD %5.2f M" % (
        k[:34], launches[k] // frames, v["SQ_INSTS_VALU"] / frames / 1e6, 100 * v["SQ_INSTS_VALU"] / frames / tot,
        v["SQ_INSTS_SALU"] / frames / 1e6, v["SQ_INSTS_LDS"] / frames / 1e6, v["SQ_INSTS_VMEM_RD"] / frames / 1e6))
blur = sum(v["SQ_INSTS_VALU"] for k, v in acc.items() if k.startswith("blur_")) / frames
print("\ntotal VALU %.1f M wave-level instructions per frame = %.2f ms of issue on 1024 SIMDs at 4 cycles each and 2.2 GHz" % (
    tot / 1e6, tot * 4 / 1024 / 2.2e9 * 1e3))
print("blur kernels: %.1f M" % (blur / 1e6))
