#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/: per-kernel statistics of a serialised 4K
# extract (one stream, so durations are not inflated by overlap), of a 1080p extract and of the
# 8192^2 frame, and the FETCH_SIZE / WRITE_SIZE passes (separate runs) that give the HBM traffic per
# kernel.  Run on the GPU box from the repository root:  bash tools/collect_profiles.sh <tag>
# Results land in gpurun_out/prof_<tag>/; copy the summaries to profiles/.
set -e
tag=${1:-r03}
root=$PWD
out=$root/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
export SIFT_HIP_SERIAL=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats4k -- python3 tools/profile_extract.py --steps 5 > $out/stats4k.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1080 -- python3 tools/profile_extract.py --steps 5 --width 1920 --height 1080 > $out/stats1080.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats8k -- python3 tools/profile_extract.py --steps 3 --width 8192 --height 8192 --octaves 6 > $out/stats8k.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats4k_fast -- python3 tools/profile_extract.py --steps 5 --blur-mode 1 > $out/stats4k_fast.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/profile_extract.py --steps 2 > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 tools/profile_extract.py --steps 2 > $out/pmc_write.log 2>&1
python3 tools/summarize_pmc.py $out/pmc_fetch $out/pmc_write 2 > $out/traffic.json
# the same two passes for the float blur (blur_mode 1), the mode the HBM roofline applies to
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_fast -- python3 tools/profile_extract.py --steps 2 --blur-mode 1 > $out/pmc_fetch_fast.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_fast -- python3 tools/profile_extract.py --steps 2 --blur-mode 1 > $out/pmc_write_fast.log 2>&1
python3 tools/summarize_pmc.py $out/pmc_fetch_fast $out/pmc_write_fast 2 > $out/traffic_fast.json
unset SIFT_HIP_SERIAL
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_issue -- python3 tools/profile_blur_level.py > $out/pmc_issue.log 2>&1
python3 tools/summarize_issue.py $out/pmc_issue > $out/issue.json
# the matcher: matrix-pipe busy cycles and executed int8 operations of one 2^20 x 2^20 launch; LDS counters in a pass of their own
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_match -- python3 tools/profile_match.py > $out/pmc_match.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_match_lds -- python3 tools/profile_match.py > $out/pmc_match_lds.log 2>&1 || true
timeout -k 10 100 python3 tools/match_rate.py 1048576 3 > $out/match_rate.log 2>&1
ms=$(python3 -c "import re,sys; t=open('$out/match_rate.log').read(); print(min(float(x) for x in re.search(r': ([0-9. ]+) ms', t).group(1).split()))")
if ls $out/pmc_match_lds/*/*counter_collection.csv > /dev/null 2>&1; then lds=$out/pmc_match_lds; else lds=""; fi
python3 tools/summarize_match_pmc.py $out/pmc_match $lds $ms > $out/match_pmc.json
for d in stats4k stats1080 stats8k stats4k_fast; do cp $out/$d/*/*kernel_stats.csv $out/${d}_kernel_stats.csv; done
rm -rf $out/stats4k $out/stats1080 $out/stats8k $out/stats4k_fast $out/pmc_fetch $out/pmc_write $out/pmc_fetch_fast $out/pmc_write_fast $out/pmc_issue $out/pmc_match $out/pmc_match_lds
ls -la $out
