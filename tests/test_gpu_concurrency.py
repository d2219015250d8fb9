"""GPU tests of rt_render's per-call scratch sets and of the limits that size them, through the C ABI.

The reference renders one camera per host thread from one BvhInterface (src/main.cpp:533-552); rt_render gives every
call a scratch set of its own, so such calls overlap on one GPU.  The sizes of a set follow the frame AND the light
sampling (a visibility byte and a work item per shadow-casting light sample of every queued hit).
"""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

import oracle_lib as o

rt = o.rt
pytestmark = pytest.mark.gpu

IMG_TOL = 2e-6


@pytest.fixture(scope="module")
def lib():
    l = rt.Library()
    l.init(0)
    return l


def _scene(lib, name):
    g = o.golden(name)
    sc = rt.SceneData.from_golden(g)
    return sc, o.OracleScene(sc), lib.bvh_build(lib.scene_create(sc))


def test_two_cameras_from_two_threads_on_one_hierarchy(lib):
    """Two host threads render two cameras from ONE rt_bvh at the same time, as the reference's per-camera threads do:
    both frames equal the ones rendered one after the other, and the two calls together take less wall time than the
    two serial calls (kernels of one call run beside the other's frame copy and host work)."""
    sc, _, bvh = _scene(lib, "teapot")
    W, H = 1920, 1080
    cams = [rt.make_camera(width=W, height=H, **o.CAM_A), rt.make_camera(width=W, height=H, **o.CAM_B)]
    feats = rt.features(shading=1, hard_shadow=1, multiple_rays=1)
    params = rt.default_params(ray_multiplier=2)
    lights, n = sc.light_array()

    def one(i, out):
        out[i] = lib.render(bvh, cams[i], lights, n, feats, params, W, H)[0]

    serial = {}
    for i in range(2):
        one(i, serial)  # (also warms both up: scratch sets, code objects)
    assert not np.array_equal(serial[0], serial[1])
    best_serial, best_parallel = float("inf"), float("inf")
    for _ in range(5):
        t0 = time.perf_counter()
        tmp = {}
        for i in range(2):
            one(i, tmp)
        best_serial = min(best_serial, time.perf_counter() - t0)
        both = {}
        threads = [threading.Thread(target=one, args=(i, both)) for i in range(2)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        best_parallel = min(best_parallel, time.perf_counter() - t0)
        for i in range(2):
            assert np.array_equal(both[i], serial[i])
    print(f"two 1080p frames: serial {best_serial * 1e3:.2f} ms, two threads {best_parallel * 1e3:.2f} ms")
    assert best_parallel < best_serial


def test_many_threads_share_the_scratch_sets(lib):
    """Twelve threads on one hierarchy: more callers than scratch sets (eight), so some wait for a set; every frame is
    the serial one.  Feature sets alternate between the queue passes, the per-lane kernel and the wavefront passes."""
    sc, _, bvh = _scene(lib, "transparent_cornell")
    W, H = 160, 96
    lights, n = sc.light_array()
    jobs = []
    for k in range(12):
        feats = [rt.features(shading=1, hard_shadow=1), rt.features(shading=1, recursive=1, transparency=1),
                 rt.features(shading=1, glossy=1, transparency=1)][k % 3]
        c = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 15.0 * k, 0.0))
        jobs.append((c, feats, rt.default_params(glossy_filter_size=8, ray_depth=2)))
    expected = [lib.render(bvh, c, lights, n, f, p, W, H)[0] for c, f, p in jobs]
    got = [None] * len(jobs)

    def work(k):
        c, f, p = jobs[k]
        for _ in range(3):
            got[k] = lib.render(bvh, c, lights, n, f, p, W, H)[0]

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(len(jobs)):
        assert np.array_equal(got[k], expected[k]), k


def test_glossy_frame_under_a_light_with_thousands_of_samples(lib):
    """A glossy frame under an area light sampled 48 x 48 times per hit: the hit queue of the wavefront passes is sized
    by the light sampling as well (entries x samples per hit stays below 2^32 and within the memory budget), so the
    frame renders, in more chunks, instead of being refused."""
    sc, orc, bvh = _scene(lib, "cornell_para")
    W, H = 24, 16
    c = rt.make_camera(width=W, height=H, **o.CAM_A)
    feats = rt.features(shading=1, soft_shadow=1, glossy=1)
    params = rt.default_params(samples_per_unit_parallel=120, glossy_filter_size=3, ray_depth=1)
    lights, n = sc.light_array()
    img, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    ref, ostats = orc.render(c, feats, params, W, H)
    # 48 x 48 samples per lit hit; the reference also lights the specular hits whose direct light render.cpp:61 then
    # drops, the wavefront passes do not queue those
    per_hit = 48 * 48
    lit = stats.shadow_rays + stats.culled_shadow_rays
    assert lit % per_hit == 0 and ostats.shadow_rays % per_hit == 0
    assert 50 * per_hit <= lit <= ostats.shadow_rays
    assert stats.bounce_rays == ostats.bounce_rays > 0
    err = np.abs(img - ref).max(axis=2)
    assert (err > 1e-4).mean() <= 0.01, f"{(err > 1e-4).sum()} pixels differ, max {err.max()}"


def test_worst_case_call_tree_beyond_the_node_buffer(lib):
    """glossy_filter_size ^ ray_depth bounds the call tree of one pixel sample: 64 samples at each of 4 depths are 17 M nodes
    at worst, more than the node buffer may hold twice over.  The frame still goes through the wavefront passes, because
    the trees that really occur are what counts (nothing in this scene is specular: every tree is its root)."""
    sc, orc, bvh = _scene(lib, "quad_textured")
    W, H = 48, 32
    c = rt.make_camera(width=W, height=H, **o.CAM_A)
    feats = rt.features(shading=1, glossy=1, texture_mapping=1, hard_shadow=1)  # (shadow rays: glossy frames take the passes)
    lights, n = sc.light_array()
    for depth in (2, 4):
        params = rt.default_params(glossy_filter_size=64, ray_depth=depth)
        img, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
        ref, _ = orc.render(c, feats, params, W, H)
        assert np.abs(img - ref).max() <= IMG_TOL
        if os.environ.get("RT_WAVEFRONT") != "0":
            assert stats.wavefront_nodes >= W * H


_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import oracle_lib as o
rt = o.rt
lib = rt.Library(); lib.init(0)
sc = rt.SceneData.from_golden(o.golden("cornell"))
bvh = lib.bvh_build(lib.scene_create(sc))
W, H = 32, 20
lights, n = sc.light_array()
img, st = lib.render(bvh, rt.make_camera(width=W, height=H, **o.CAM_A), lights, n, rt.features(shading=1, glossy=1, hard_shadow=1),
                     rt.default_params(glossy_filter_size=512, ray_depth=1), W, H)
np.save(sys.argv[2], img)
print("RAYS", st.primary_rays, st.bounce_rays, st.shadow_rays + st.culled_shadow_rays, "NODES", st.wavefront_nodes)
"""


def test_a_single_sample_that_overflows_the_node_buffer_goes_to_the_per_lane_kernel(tmp_path):
    """With a node buffer of 256 nodes (RT_WAVEFRONT_NODES) the call tree of one pixel sample of a 512-sample glossy frame
    cannot fit, however small the chunk: the pass is handed to the per-lane kernel.  Frame and ray counts are those of a
    run that never tried the wavefront passes (nothing of the abandoned attempts stays in the counters)."""
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for name, env in (("fallback", dict(RT_WAVEFRONT="1", RT_WAVEFRONT_NODES="256")), ("per_lane", dict(RT_WAVEFRONT="0"))):
        path = str(tmp_path / (name + ".npy"))
        run = subprocess.run([sys.executable, "-c", _CHILD, here, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
        outs.append((np.load(path), [l for l in run.stdout.splitlines() if l.startswith("RAYS")][-1]))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1], (outs[0][1], outs[1][1])
    assert outs[0][1].endswith("NODES 0")
