#!/bin/bash
# GPU parity tests, then a short bench; prints the per-pass times.  Run through gpurun from the repo root.
set -u
TAG=${1:-chk}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/$TAG.test.log 2>&1
rc=$?
tail -3 gpurun_out/$TAG.test.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/$TAG.bench.log 2>&1 || { tail -5 gpurun_out/$TAG.bench.log; exit 1; }
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$TAG.bench.log") if l.startswith("{")][-1])
a=d["roofline"]["all_kernels"]
print("frame_ms %.3f  primary %.3f cull %.3f shadow %.3f shade %.3f" % (d["ms_per_step"], a["primary_ms"], a["cull_ms"], a["shadow_ms"], a["shade_ms"]))
print(d["roofline"].get("stream_steps"))
PY
