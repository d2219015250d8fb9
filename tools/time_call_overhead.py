"""Wall time of one rt_render call against the device time it reports, on a small frame (Cornell box 512 x 512, mirror
reflections) and on an eighth of the benchmark frame: what the host side of a call costs.  python tools/time_call_overhead.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as o
rt = o.rt
synth = importlib.import_module("ray-tracer_amd.synth")
lib = rt.Library(); lib.init(0)

def measure(name, call):
    for _ in range(5): call()
    walls, devs = [], []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = call()
        walls.append((time.perf_counter() - t0) * 1e3); devs.append(st.kernel_ms + st.resolve_ms)
    walls.sort(); devs.sort()
    print(f"{name}: wall {walls[15]:.3f} ms, device {devs[15]:.3f} ms, host side {walls[15] - devs[15]:.3f} ms", flush=True)

sc = rt.SceneData.from_golden(o.golden("cornell"))
bvh = lib.bvh_build(lib.scene_create(sc))
lights, n = sc.light_array()
W = H = 512
out = torch.empty(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
cam = rt.make_camera(width=W, height=H, **o.CAM_A)
feats = rt.features(shading=1, recursive=1)
params = rt.default_params()
measure("cornell 512x512 mirror", lambda: lib.render(bvh, cam, lights, n, feats, params, W, H, 0, H, out=out.data_ptr(), device_ptr=1, stream=0)[1])
feats0 = rt.features(shading=1)
measure("cornell 512x512 shading only", lambda: lib.render(bvh, cam, lights, n, feats0, params, W, H, 0, H, out=out.data_ptr(), device_ptr=1, stream=0)[1])

import bench
scene = rt.SceneData([synth.dragon_class_mesh(6)], lights=synth.area_light())
bvh2 = lib.bvh_build(lib.scene_create(scene))
l2, n2 = scene.light_array()
cam2 = rt.make_camera(width=bench.WIDTH, height=bench.HEIGHT, **bench.CAMERA)
f2 = rt.features(shading=1, soft_shadow=1, multiple_rays=1)
p2 = rt.default_params(ray_multiplier=4)
measure("benchmark frame, share 0 of 8", lambda: lib.render_blocks(bvh2, cam2, l2, n2, f2, p2, bench.WIDTH, bench.HEIGHT, 16, 0, 8, out=out.data_ptr(), device_ptr=1, stream=0)[1])
measure("benchmark frame, whole", lambda: lib.render(bvh2, cam2, l2, n2, f2, p2, bench.WIDTH, bench.HEIGHT, 0, bench.HEIGHT, out=out.data_ptr(), device_ptr=1, stream=0)[1])
