#!/bin/bash
# Per-kernel durations of a serialised extract (one stream, so durations are not inflated by overlap):
#   bash tools/kernel_trace.sh [profile_extract.py arguments]      (on the GPU box, from the repository root)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$OLDPWD}
export SIFT_HIP_SERIAL=1
rm -rf gpurun_out/kt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 tools/profile_extract.py --steps 5 "$@" > gpurun_out/kt.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void sifthip::", "").replace("sifthip::", "")
    print("%-40s calls/frame %5.1f  avg %8.1f us  per frame %8.1f us" % (n[:40], int(r["Calls"]) / 5, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 5e3))
PY
