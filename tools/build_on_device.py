"""Builds the 1,310,720-triangle mesh's hierarchy on the host and (three times) on the GPU and prints the times; the
commands profiles/r3_build and profiles/r3_build_sah were recorded from:  python3 tools/build_on_device.py [level] [sah]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd"); synth = importlib.import_module("ray-tracer_amd.synth")
lib = rt.Library(); lib.init(0)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sah = len(sys.argv) > 2 and sys.argv[2] == "sah"
sc = rt.SceneData([synth.dragon_class_mesh(level)], lights=synth.area_light())
sh = lib.scene_create(sc)
t0 = time.perf_counter(); host = lib.bvh_build(sh, sah=sah); t1 = time.perf_counter()
print(f"{sc.n_triangles} triangles{', SAH with 64 bins' if sah else ''}: host build {t1 - t0:.3f} s", flush=True)
import numpy as np
hn, hp = lib.bvh_export(host)
for i in range(3):
    t0 = time.perf_counter(); dev, on = lib.bvh_build_device(sh, sah=sah); t1 = time.perf_counter()
    print(f"  rt_bvh_build_device call {i}: {t1 - t0:.4f} s (on the GPU: {on})", flush=True)
    if i == 0:
        dn, dp = lib.bvh_export(dev)
        print(f"  same tree as the host's: {np.array_equal(hn, dn) and np.array_equal(hp, dp)} ({len(hn)} nodes, {lib.lib.rt_bvh_num_levels(dev)} levels)", flush=True)
    lib.lib.rt_bvh_destroy(dev)
