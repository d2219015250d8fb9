"""Times, on ONE GPU, the interleaved row-block share of every rank of a 1/2/4/8-rank job on the benchmark frame
(rt_render_blocks): the slowest share bounds the multi-GPU frame time.  python tools/time_rank_shares.py [world]
"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("ray-tracer_amd"); synth = importlib.import_module("ray-tracer_amd.synth")
lib = rt.Library(); lib.init(0)
scene = rt.SceneData([synth.dragon_class_mesh(6)], lights=synth.area_light())
bvh = lib.bvh_build(lib.scene_create(scene))
lights, n = scene.light_array()
W, H = 1920, 1080
RB = int(os.environ.get("RB", "16"))
cam = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0, 0, 0), rot_deg=(20.0, 20.0, 0.0))
feats = rt.features(shading=1, soft_shadow=1, multiple_rays=1)
p = rt.default_params(ray_multiplier=4)
out = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
for world in ([int(sys.argv[1])] if len(sys.argv) > 1 else (1, 2, 4, 8)):
    tot = 0.0
    for r in range(world):
        for i in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, st = lib.render_blocks(bvh, cam, lights, n, feats, p, W, H, RB, r, world, out=out.data_ptr(), device_ptr=1, stream=0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        tot = max(tot, dt)
        last = (dt, st.kernel_ms, st.primary_ms, st.shadow_ms)
    print(f"world {world}: slowest rank wall {tot:.3f} ms; last rank wall/kernel/primary/shadow {last}", flush=True)
