"""Container-only: the compiled reference (oracle/_ref/ref_dump --time-obj, one thread) against the oracle (one thread)
on the BENCHMARK's own mesh — the 81,920-triangle dragon-class mesh under the benchmark's parallelogram light, shading
+ soft shadows (16 light samples), default camera — at a frame small enough for the reference to finish (it copies the
whole Mesh per leaf primitive it tests, bounding_volume_hierarchy.cpp:339).  The ratio translates bench.py's
cpu_baseline (kind "port": the oracle) into "the reference as it is" at the benchmark's mesh size.
Needs /root/reference; run: python tools/time_reference_headline.py [level] [width] [height] [n]"""
import importlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o
rt = o.rt
synth = importlib.import_module("ray-tracer_amd.synth")
level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
W = int(sys.argv[2]) if len(sys.argv) > 2 else 48
H = int(sys.argv[3]) if len(sys.argv) > 3 else 27
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1
mesh = synth.dragon_class_mesh(level)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "mesh.obj")
    with open(path, "w") as f:  # nine significant digits: every float32 survives the decimal round trip
        for v in mesh["vertices"]:
            f.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
        for v in mesh["vertices"]:
            f.write("vn %.9g %.9g %.9g\n" % (v[3], v[4], v[5]))
        for t in mesh["triangles"]:
            f.write("f %d//%d %d//%d %d//%d\n" % (t[0] + 1, t[0] + 1, t[1] + 1, t[1] + 1, t[2] + 1, t[2] + 1))
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_dump"), "--time-obj", path, str(W), str(H), str(n)], capture_output=True, text=True, check=True).stdout
tris, _, _, _, ms = out.split()[-5:]
sc = rt.SceneData([mesh], lights=synth.area_light())
orc = o.OracleScene(sc)
feats = rt.features(shading=1, soft_shadow=1, multiple_rays=int(n > 1))
params = rt.default_params(ray_multiplier=n)
cam = rt.make_camera(width=W, height=H, **o.CAM_A)
best = None
for _ in range(3):
    t0 = time.perf_counter()
    _, st = orc.render(cam, feats, params, W, H, threads=1)
    dt = (time.perf_counter() - t0) * 1e3
    best = dt if best is None else min(best, dt)
print(f"{tris} triangles, {W}x{H}, {n*n} spp, {st.total_rays()} rays: compiled reference {float(ms):.0f} ms, oracle {best:.1f} ms, ratio {float(ms)/best:.0f}")
