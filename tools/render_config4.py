"""Renders Config 4 (1920x1080, 36 spp, transparency + glossy reflection with 64 samples per bounce) three times on one of
its scenes and prints the device time: python tools/render_config4.py [transparent_cornell|transparent_cubes]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as o
rt = o.rt
lib = rt.Library(); lib.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "transparent_cornell"
sc = rt.SceneData.from_golden(o.golden(which))
bvh = lib.bvh_build(lib.scene_create(sc)); lights, n = sc.light_array()
W, H = 1920, 1080
out = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
cam = rt.make_camera(width=W, height=H, **o.CAM_A)
feats = rt.features(shading=1, transparency=1, glossy=1, multiple_rays=1)
params = rt.default_params(ray_multiplier=6, glossy_filter_size=64)
for i in range(3):
    _, st = lib.render(bvh, cam, lights, n, feats, params, W, H, 0, H, out=out.data_ptr(), device_ptr=1, stream=0)
    print(which, "kernel ms", st.kernel_ms, "nodes", st.wavefront_nodes, "rays", st.total_rays(), flush=True)
