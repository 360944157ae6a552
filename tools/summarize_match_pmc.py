#!/usr/bin/env python3
"""Turns a rocprofv3 --pmc run of tools/profile_match.py (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE, optionally a second directory with LDS counters) into the
figures kept under profiles/:  python tools/summarize_match_pmc.py <dir> [<lds dir>] [ms of the launch] > profiles/rNN_match_pmc.json
The last match_kernel dispatch of the run is the one reported (the first is the warm-up)."""
import csv
import glob
import hashlib
import json
import os
import sys


def last_dispatch(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "match_kernel" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    return {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last}, rows[0]["Kernel_Name"].split("(")[0]


args = [a for a in sys.argv[1:]]
ms = float(args.pop()) if len(args) > 1 and args[-1].replace(".", "").isdigit() else None
raw, name = last_dispatch(args[0])
lds = last_dispatch(args[1])[0] if len(args) > 1 else {}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 1 << 20
out = {"workload": "%s, %d x %d descriptors (tools/profile_match.py), one launch" % (name.replace("void sifthip::", ""), n, n),
       "kernel_source_sha256": hashlib.sha256(open(os.path.join(root, "sift_amd", "csrc", "match.hip"), "rb").read()).hexdigest(),
       "raw": raw, "raw_lds_pass": lds, "derived": {}}
d = out["derived"]
simds = 1024.0
if "GRBM_GUI_ACTIVE" in raw:
    d["kernel_cycles_per_xcd"] = raw["GRBM_GUI_ACTIVE"] / 8
if "SQ_VALU_MFMA_BUSY_CYCLES" in raw and "GRBM_GUI_ACTIVE" in raw:
    d["mfma_busy_cycles_per_simd"] = raw["SQ_VALU_MFMA_BUSY_CYCLES"] / simds
    d["mfma_pipe_busy_fraction"] = round(d["mfma_busy_cycles_per_simd"] / d["kernel_cycles_per_xcd"], 4)
if "SQ_INSTS_VALU_MFMA_MOPS_I8" in raw:
    d["int8_ops_executed"] = raw["SQ_INSTS_VALU_MFMA_MOPS_I8"] * 512
    d["int8_ops_algorithmic"] = float(n) * n * 256
    d["executed_over_algorithmic"] = round(d["int8_ops_executed"] / d["int8_ops_algorithmic"], 4)
if ms and "kernel_cycles_per_xcd" in d:
    d["launch_ms"] = ms
    d["clock_GHz_under_load"] = round(d["kernel_cycles_per_xcd"] / (ms * 1e-3) / 1e9, 3)
if lds:
    d["lds"] = {k: v for k, v in lds.items()}
    if lds.get("SQ_LDS_BANK_CONFLICT") is not None and lds.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction_of_active"] = round(lds["SQ_LDS_BANK_CONFLICT"] / lds["SQ_LDS_IDX_ACTIVE"], 4)
d["note"] = ("MOPS counts units of 512 int8 operations; one v_mfma_i32_32x32x32_i8 = 65536 operations = 32 busy cycles, "
             "i.e. 2048 ops per cycle per SIMD = 5.0 POP/s at 1024 SIMDs x 2.4 GHz")
print(json.dumps(out, indent=1))
