#!/bin/bash
# usage: sweep_variants.sh "<env settings>" name...   one short bench per build under build_variants/ ("product" = the in-tree library)
settings=$1; shift
for v in "$@"; do
  lib=$PWD/build_variants/librt_amd_$v.so
  [ "$v" = product ] && lib=$PWD/ray-tracer_amd/csrc/librt_amd.so
  env $settings RT_AMD_LIBRARY=$lib timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra $BENCH_ARGS > gpurun_out/sweep.log 2>&1 || { echo "$v: FAILED"; tail -3 gpurun_out/sweep.log; exit 1; }
  python3 - "$v $settings" <<PY
import json,sys
d=json.loads([l for l in open("gpurun_out/sweep.log") if l.startswith("{")][-1])
a=d["roofline"]["all_kernels"]; s=d["roofline"]["stream_steps"]
print("%-40s frame %.3f primary %.3f cull %.3f shadow %.3f shade %.3f | box steps %.2fM prim steps %.2fM entries/step %.1f" % (sys.argv[1], d["ms_per_step"], a["primary_ms"], a["cull_ms"], a["shadow_ms"], a["shade_ms"], s["box_steps"]/1e6, s["primitive_steps"]/1e6, s["entries_per_box_step"]), flush=True)
PY
done
