#!/bin/bash
# Kernel trace of the general per-lane kernel k_render on the bounce configurations (Config 1: Cornell 512x512 mirror
# recursion; Config 4 with the wavefront passes switched off), from tools/time_configs.py:
#   profiles/run_profile_render.sh   (on the GPU box, through gpurun)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp RT_WAVEFRONT=0
OUT=gpurun_out/prof_r2_render
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_configs.py > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/prof_r2_render/summary.txt
import csv, glob
print("== tools/time_configs.py with RT_WAVEFRONT=0 (every bounce frame in k_render): device times it prints ==")
for line in open("gpurun_out/prof_r2_render/log.txt"):
    if ": kernel " in line: print(line.rstrip())
print("== rocprofv3 --kernel-trace --stats of the same run (4 frames per configuration) ==")
for f in glob.glob("gpurun_out/prof_r2_render/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_" in r["Name"]:
            print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.3f}")
PY
cat gpurun_out/prof_r2_render/summary.txt
