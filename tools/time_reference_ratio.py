"""Container-only: the compiled reference (oracle/_ref/ref_dump --time, one thread) against the oracle (one thread) on
the same frames.  The ratio translates the benchmark's CPU baseline (kind "port": the oracle) into "the reference as
it is".  Needs /root/reference; run: python tools/time_reference_ratio.py"""
import importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o
rt = o.rt
out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_dump"), "--time"], capture_output=True, text=True, check=True).stdout
for line in (l for l in out.splitlines() if len(l.split()) == 5 and l.split()[1].isdigit()):
    name, W, H, n, ms = line.split()
    W, H, n, ms = int(W), int(H), int(n), float(ms)
    scene = name.replace("_4spp", "")
    orc = o.OracleScene(rt.SceneData.from_golden(o.golden(scene)))
    feats = rt.features(shading=1, soft_shadow=int(scene == "cornell_para"), multiple_rays=int(n > 1))
    params = rt.default_params(ray_multiplier=n)
    cam = rt.make_camera(width=W, height=H, **o.CAM_A)
    t0 = time.perf_counter()
    _, st = orc.render(cam, feats, params, W, H, threads=1)
    oms = (time.perf_counter() - t0) * 1e3
    print(f"{name}: {W}x{H}, {n*n} spp, {st.total_rays()} rays: compiled reference {ms:.0f} ms, oracle {oms:.0f} ms, ratio {ms/oms:.1f}")
