"""Times rt_render on the other benchmark configurations at full size on one GPU (no parity check here; these are
the numbers the design notes quote).  Run on a GPU box: python tools/time_configs.py
"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as o
rt = o.rt
lib = rt.Library(); lib.init(0)

def run(name, scene, W, H, feats, params, cam):
    sc = rt.SceneData.from_golden(o.golden(scene))
    bvh = lib.bvh_build(lib.scene_create(sc))
    lights, n = sc.light_array()
    out = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
    best = None
    for i in range(4):
        _, st = lib.render(bvh, cam, lights, n, feats, params, W, H, 0, H, out=out.data_ptr(), device_ptr=1, stream=0)
        best = st.kernel_ms if best is None else min(best, st.kernel_ms)
    print(f"{name}: kernel {best:.3f} ms, rays {st.total_rays()/1e6:.2f} M, {st.total_rays()/best/1e3:.0f} Mrays/s, wavefront nodes {st.wavefront_nodes/1e6:.2f} M", flush=True)

A = o.CAM_A
run("C1 cornell 512^2 mirror depth 4", "cornell", 512, 512, rt.features(shading=1, recursive=1), rt.default_params(), rt.make_camera(width=512, height=512, **A))
run("C2 monkey 1080p 4spp envmap", "monkey", 1920, 1080, rt.features(shading=1, texture_mapping=1, environment_mapping=1, multiple_rays=1), rt.default_params(ray_multiplier=2, enabled_skybox=1), rt.make_camera(width=1920, height=1080, **A))
run("C2 cube_textured 1080p 4spp envmap", "cube_textured", 1920, 1080, rt.features(shading=1, texture_mapping=1, environment_mapping=1, multiple_rays=1), rt.default_params(ray_multiplier=2, enabled_skybox=1), rt.make_camera(width=1920, height=1080, **A))
for sc in ("transparent_cornell", "transparent_cubes"):
    run(f"C4 {sc} 1080p 36spp transparency+glossy64", sc, 1920, 1080, rt.features(shading=1, transparency=1, glossy=1, multiple_rays=1), rt.default_params(ray_multiplier=6, glossy_filter_size=64), rt.make_camera(width=1920, height=1080, **A))
run("teapot 1080p 16spp soft", "teapot", 1920, 1080, rt.features(shading=1, soft_shadow=1, multiple_rays=1), rt.default_params(ray_multiplier=4), rt.make_camera(width=1920, height=1080, **A))

