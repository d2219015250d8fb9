#!/bin/bash
# Kernel trace of Config 4 (1920x1080, 36 spp, transparency + glossy reflection with 64 samples per bounce) on both of its
# scenes, wavefront passes (the default for glossy frames):  profiles/run_profile_c4.sh  (on the GPU box, through gpurun)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r2_c4
rm -rf $OUT; mkdir -p $OUT
for scene in transparent_cornell transparent_cubes; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$scene -- python3 tools/render_config4.py $scene > $OUT/$scene.log 2>&1 || exit 1
done
python3 - <<'PY' > gpurun_out/prof_r2_c4/summary.txt
import csv, glob
for scene in ("transparent_cornell", "transparent_cubes"):
    print(f"== {scene}: rocprofv3 --kernel-trace --stats of 3 frames (tools/render_config4.py) ==")
    for line in open(f"gpurun_out/prof_r2_c4/{scene}.log"):
        if line.startswith(scene): print(line.rstrip())
    for f in glob.glob(f"gpurun_out/prof_r2_c4/{scene}/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} {float(r['Percentage']):6.2f}%")
PY
cat gpurun_out/prof_r2_c4/summary.txt
