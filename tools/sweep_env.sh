#!/bin/bash
# usage: sweep_env.sh "VAR=a VAR2=b" "VAR=c" ...   one short bench per environment setting; prints per-pass times
for setting in "$@"; do
  env $setting timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/sweep.log 2>&1 || { echo "$setting: FAILED"; tail -3 gpurun_out/sweep.log; exit 1; }
  python3 - "$setting" <<PY
import json,sys
d=json.loads([l for l in open("gpurun_out/sweep.log") if l.startswith("{")][-1])
a=d["roofline"]["all_kernels"]
print("%-40s frame %.3f primary %.3f cull %.3f shadow %.3f shade %.3f" % (sys.argv[1], d["ms_per_step"], a["primary_ms"], a["cull_ms"], a["shadow_ms"], a["shade_ms"]), flush=True)
PY
done
