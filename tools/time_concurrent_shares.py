"""The headline frame as one rt_render call against the same frame as K concurrent rt_render_blocks calls (interleaved
16-row blocks, a host thread and a stream each) on one GPU: do the stages of different shares overlap usefully?
python3 tools/time_concurrent_shares.py [K]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rt = importlib.import_module("ray-tracer_amd"); synth = importlib.import_module("ray-tracer_amd.synth")
import bench
lib = rt.Library(); lib.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H = bench.WIDTH, bench.HEIGHT
sc = rt.SceneData([synth.dragon_class_mesh(6)], lights=synth.area_light())
sh = lib.scene_create(sc)
bvh, _ = lib.bvh_build_device(sh)
lights, n = sc.light_array()
cam = rt.make_camera(width=W, height=H, **bench.CAMERA)
feats = rt.features(shading=1, soft_shadow=1, multiple_rays=1)
params = rt.default_params(ray_multiplier=4)
dev = torch.device("cuda", 0)
full = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
shares = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
s0 = torch.cuda.current_stream().cuda_stream

def one():
    t0 = time.perf_counter()
    _, st = lib.render(bvh, cam, lights, n, feats, params, W, H, 0, H, out=full.data_ptr(), device_ptr=1, stream=s0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, st.kernel_ms + st.resolve_ms

def many():
    def work(k):
        lib.render_blocks(bvh, cam, lights, n, feats, params, W, H, 16, k, K, out=shares[k].data_ptr(), device_ptr=1, stream=streams[k].cuda_stream)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

for _ in range(3): one(); many()
a = sorted(one() for _ in range(9))[4]
b = sorted(many() for _ in range(9))[4]
print(f"one call: wall {a[0]:.2f} ms (device {a[1]:.2f} ms); {K} concurrent shares: wall {b:.2f} ms", flush=True)
