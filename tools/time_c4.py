"""Times Config 4 (1920x1080, 36 spp, transparency + glossy reflection with 64 samples) on both scenes: python tools/time_c4.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as o
rt = o.rt
lib = rt.Library(); lib.init(0)
W, H = 1920, 1080
for scene in ("transparent_cornell", "transparent_cubes"):
    sc = rt.SceneData.from_golden(o.golden(scene))
    bvh = lib.bvh_build(lib.scene_create(sc))
    lights, n = sc.light_array()
    out = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
    feats = rt.features(shading=1, transparency=1, glossy=1, multiple_rays=1)
    params = rt.default_params(ray_multiplier=6, glossy_filter_size=64)
    cam = rt.make_camera(width=W, height=H, **o.CAM_A)
    best = None
    for i in range(5):
        _, st = lib.render(bvh, cam, lights, n, feats, params, W, H, 0, H, out=out.data_ptr(), device_ptr=1, stream=0)
        best = st.kernel_ms if best is None else min(best, st.kernel_ms)
    print(f"{scene}: kernel {best:.3f} ms, rays {st.total_rays()/1e6:.2f} M, wavefront nodes {st.wavefront_nodes/1e6:.2f} M", flush=True)
