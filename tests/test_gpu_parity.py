"""GPU parity tests: the HIP path, called through the C ABI (librt_amd.so), against the oracle on the same
inputs and against the committed golden vectors of the compiled reference.

Bar (integer / index results bit-exact; floating point as stated):
  - hit flags, mesh and triangle ids, t, barycentrics, normals: bit-exact;
  - shadow visibility bits: bit-exact;
  - images: every channel within 2e-6 of the oracle (the kernels evaluate the same fp32 operation sequence;
    the only divergence allowed is the last-ulp difference of device pow / log / cos against glibc's),
    which is far inside the north star's 1/255 on 99.9 % of pixels.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as o

rt = o.rt
pytestmark = pytest.mark.gpu

IMG_TOL = 2e-6
_state = {}


@pytest.fixture(scope="module")
def lib():
    l = rt.Library()
    l.init(0)
    return l


def setup(lib, name):
    if name not in _state:
        g = o.golden(name)
        sc = rt.SceneData.from_golden(g)
        sh = lib.scene_create(sc)
        _state[name] = (g, sc, o.OracleScene(sc), sh, lib.bvh_build(sh))
    return _state[name]


def cam(which, size):
    spec = o.CAM_A if which == "A" else o.CAM_B
    return rt.make_camera(width=size[0], height=size[1], **spec)


def size_of(which):
    return o.SIZE_A if which == "A" else o.SIZE_B


def _hit_cases():
    out = []
    for s in gc.SCENES:
        for mode in ("prune", "noprune", "brute"):
            if s == "teapot" and mode == "brute":
                continue
            out.append((s, "A", mode))
        if s in gc.B_SCENES:
            out.append((s, "B", "prune"))
            out.append((s, "B", "noprune"))
    return out


@pytest.mark.parametrize("scene,which,mode", _hit_cases())
def test_hit_records_bit_exact(lib, scene, which, mode):
    g, sc, orc, sh, bvh = setup(lib, scene)
    W, H = size_of(which)
    rays = o.camera_rays(cam(which, (W, H)), W, H)
    feats = {"prune": rt.features(normal_interp=1, texture_mapping=1), "noprune": rt.features(recursive=1),
             "brute": rt.features(normal_interp=1, texture_mapping=1, accel_structure=0)}[mode]
    h = lib.intersect(bvh, rays, feats)
    p = f"{which}.{mode}."
    # against the compiled reference
    assert np.array_equal(h["hit"], g[p + "hit"])
    assert np.array_equal(h["t"], g[p + "t"])
    m = h["hit"] == 1
    if mode != "brute":
        assert np.array_equal(h["mesh"][m], g[p + "mesh"][m])
        tri_known = m & (g[p + "mesh"] >= 0)  # the reference does not expose which sphere was hit
        assert np.array_equal(h["triangle"][tri_known], g[p + "tri"][tri_known])
    if p + "normal" in g:
        assert np.array_equal(h["normal"][m], g[p + "normal"][m])
        assert np.array_equal(h["bary"][m], g[p + "bary"][m])
        assert np.array_equal(h["kd"][m], g[p + "kd"][m])
    # and against the oracle (covers the brute-force ids too)
    ho = orc.intersect(rays, feats)
    for k in ("hit", "t"):
        assert np.array_equal(h[k], ho[k])
    for k in ("mesh", "triangle", "normal", "bary", "kd"):
        assert np.array_equal(h[k][m], ho[k][m]), k


@pytest.mark.parametrize("scene", gc.SCENES)
@pytest.mark.parametrize("mode", ["prune", "noprune"])
def test_shadow_visibility_bit_exact(lib, scene, mode):
    g, sc, orc, sh, bvh = setup(lib, scene)
    W, H = o.SIZE_A
    rays = o.camera_rays(cam("A", (W, H)), W, H)
    feats = rt.features(shading=1, hard_shadow=1, soft_shadow=1, recursive=int(mode == "noprune"))
    h = lib.intersect(bvh, rays, feats)
    m = h["hit"] == 1
    hit_rays = rays[m].copy()
    hit_rays["t"] = h["t"][m]
    vis = lib.visibility(bvh, hit_rays, g[f"A.{mode}.vis_samples"], feats)
    assert np.array_equal(vis, g[f"A.{mode}.vis"][m])


def _image_cases():
    return [(s, t) for s in gc.SCENES for t in gc.image_tags(s)]


@pytest.mark.parametrize("scene,tag", _image_cases())
def test_images_match_reference(lib, scene, tag):
    g, sc, orc, sh, bvh = setup(lib, scene)
    which, feats, params = gc.IMAGE_CASES[tag]
    W, H = size_of(which)
    lights, n = sc.light_array()
    img, stats = lib.render(bvh, cam(which, (W, H)), lights, n, feats, params, W, H)
    if feats.bloom:
        img = lib.bloom(img, params)
    ref = g["img." + tag]
    assert img.shape == ref.shape
    err = np.abs(img - ref)
    assert err.max() <= IMG_TOL, f"max abs diff {err.max()} on {(err > IMG_TOL).any(axis=2).sum()} pixels"
    assert stats.primary_rays >= W * H


@pytest.mark.parametrize("scene", gc.GLOSSY_SCENES)
def test_glossy_images_match_reference(lib, scene):
    g, sc, orc, sh, bvh = setup(lib, scene)
    lights, n = sc.light_array()
    for tag, (which, (W, H), feats, params) in gc.GLOSSY_CASES.items():
        if "img." + tag not in g:
            continue
        img, _ = lib.render(bvh, cam(which, (W, H)), lights, n, feats, params, W, H)
        ref = g["img." + tag]
        # device log/cos/sqrt in the Gaussian deviates may differ from glibc in the last place; a deviate
        # that moves a perturbed ray across a triangle edge changes that sample, so allow a few pixels
        err = np.abs(img - ref).max(axis=2)
        assert (err > 1e-4).mean() <= 0.01, f"{tag}: {(err > 1e-4).sum()} pixels differ, max {err.max()}"
        assert gc.psnr(img, ref) >= 45.0, f"{tag}: PSNR {gc.psnr(img, ref):.1f} dB"


@pytest.mark.parametrize("scene", ["cube_textured", "quad_textured"])
def test_bilinear_texture_lookup_matches_oracle(lib, scene):
    """Bilinear filtering: no reference behaviour exists (its acquireTexel ignores the switch, texture.cpp:5-17), so
    this is GPU against the oracle's definition only — parity unpinned against the reference.  The filtered kd of
    every hit is bit-exact (same fp32 operations), images within the usual tolerance, and the switch changes the
    picture."""
    g, sc, orc, sh, bvh = setup(lib, scene)
    for which in ("A", "B"):
        W, H = size_of(which)
        c = cam(which, (W, H))
        rays = o.camera_rays(c, W, H)
        feats = rt.features(normal_interp=1, texture_mapping=1, bilinear_filtering=1)
        h, ho = lib.intersect(bvh, rays, feats), orc.intersect(rays, feats)
        m = h["hit"] == 1
        assert m.any() and np.array_equal(h["hit"], ho["hit"])
        assert np.array_equal(h["kd"][m], ho["kd"][m])
        nearest = lib.intersect(bvh, rays, rt.features(normal_interp=1, texture_mapping=1))
        assert not np.array_equal(nearest["kd"][m], h["kd"][m])
        lights, n = sc.light_array()
        feats = rt.features(shading=1, texture_mapping=1, bilinear_filtering=1, multiple_rays=1)
        params = rt.default_params(ray_multiplier=2)
        img, _ = lib.render(bvh, c, lights, n, feats, params, W, H)
        ref, _ = orc.render(c, feats, params, W, H)
        assert np.abs(img - ref).max() <= IMG_TOL


def _config_cases():
    return [(s, t) for s in gc.C2_SCENES for t in ("A.c2", "B.c2")] + [(s, "A.c4") for s in gc.C4_SCENES]


@pytest.mark.parametrize("scene,tag", _config_cases())
def test_config_feature_sets_match_reference(lib, scene, tag):
    """Configs 2 and 4 at fixture size against the compiled reference's frames: sky box + textures + 4 spp within the
    image tolerance; transparency + glossy (64 samples per bounce) + 4 spp to the glossy bar (device log / cos / sqrt
    in the Gaussian deviates may differ from glibc in the last place and move a sample across an edge) and
    PSNR >= 45 dB."""
    g, sc, orc, sh, bvh = setup(lib, scene)
    which, size, feats, params = gc.CONFIG_CASES[tag]
    W, H = size or size_of(which)
    lights, n = sc.light_array()
    img, stats = lib.render(bvh, cam(which, (W, H)), lights, n, feats, params, W, H)
    ref = o.golden("configs")[f"{scene}.img.{tag}"]
    err = np.abs(img - ref).max(axis=2)
    if feats.glossy:
        assert (err > 1e-4).mean() <= 0.01, f"{(err > 1e-4).sum()} pixels differ, max {err.max()}"
        assert gc.psnr(img, ref) >= 45.0
        assert stats.bounce_rays > 64 * 4  # some pixel spawned its 64 glossy rays per sample
        assert stats.transparency_overflows == 0
    else:
        assert err.max() <= IMG_TOL, f"max abs diff {err.max()} on {(err > IMG_TOL).sum()} pixels"


@pytest.mark.parametrize("scene", ["cornell_para", "monkey"])
def test_glossy_with_jittered_lights_matches_oracle(lib, scene):
    """Glossy fan-out together with jittered area lights: every call's light jitter sits at the position on the rand()
    stream that the reference's depth-first recursion would give it (after the draws of its ancestors and of its
    earlier siblings' whole subtrees), whichever order the calls are evaluated in.  The monkey (specular everywhere,
    concave, here under an area light) makes glossy calls nest and siblings' subtrees differ in size.
    Oracle only: the compiled reference has no fixture for this combination."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    g, sc, orc, sh, bvh = setup(lib, scene)
    light_rows = synth.area_light() if scene == "monkey" else None
    lights, n = sc.light_array(light_rows)
    W, H = 48, 32
    shadowed = 0
    for depth, feats in ((1, rt.features(shading=1, soft_shadow=1, glossy=1)), (2, rt.features(shading=1, soft_shadow=1, glossy=1, transparency=1, multiple_rays=1))):
        params = rt.default_params(glossy_filter_size=6, ray_multiplier=2, ray_depth=depth)
        img, stats = lib.render(bvh, cam("A", (W, H)), lights, n, feats, params, W, H)
        ref, ostats = orc.render(cam("A", (W, H)), feats, params, W, H, lights=light_rows)
        err = np.abs(img - ref).max(axis=2)
        assert (err > 1e-4).mean() <= 0.01, f"{(err > 1e-4).sum()} pixels differ, max {err.max()}"
        assert gc.psnr(img, ref) >= 45.0
        assert stats.bounce_rays == ostats.bounce_rays and stats.bounce_rays > 0
        # (the reference also lights the glossy hits and then drops that light; the kernels skip it and its shadow rays)
        assert stats.shadow_rays + stats.culled_shadow_rays <= ostats.shadow_rays
        shadowed += stats.shadow_rays
    assert shadowed > 0


def test_config5_with_glossy_reflection_matches_oracle(lib):
    """Config 5's full feature set at a size the oracle finishes: specular mesh, 16-sample soft shadows, glossy reflection
    (nested: the mesh is specular everywhere), depth of field, bloom — lens rays decided by the centre rays' pass, glossy
    calls evaluated as wavefront passes (or in the per-lane kernel when those are switched off), light jitter at the
    calls' depth-first positions on the rand() stream."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    sc = rt.SceneData([synth.dragon_class_mesh(3)], lights=synth.area_light())
    orc = o.OracleScene(sc)
    bvh = lib.bvh_build(lib.scene_create(sc))
    lights, n = sc.light_array()
    W, H = 56, 36
    c = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0))
    feats = rt.features(shading=1, soft_shadow=1, glossy=1, depth_of_field=1, bloom=1)
    params = rt.default_params(dof_samples=5, dof_f=2.5, dof=0.05, glossy_filter_size=6, ray_depth=2, bloom_threshold=0.2)
    img, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    ref, ostats = orc.render(c, feats, params, W, H)
    err = np.abs(img - ref).max(axis=2)
    assert (err > 1e-4).mean() <= 0.01, f"{(err > 1e-4).sum()} pixels differ, max {err.max()}"
    assert gc.psnr(img, ref) >= 45.0
    assert stats.primary_rays == ostats.primary_rays and stats.bounce_rays == ostats.bounce_rays and stats.bounce_rays > 0
    assert np.abs(lib.bloom(img.copy(), params) - o.bloom(ref, params)).max() <= 1e-4


def _many_lights(count, seed=5):
    """`count` lights above the scene: point, segment and parallelogram lights in turn, dim enough to keep the sum in range."""
    rng = np.random.default_rng(seed)
    rows = np.zeros((count, 22), np.float32)
    for i in range(count):
        pos = np.array([rng.uniform(-0.6, 0.6), rng.uniform(0.2, 0.55), rng.uniform(-0.6, 0.6)], np.float32)
        col = (rng.uniform(0.2, 1.0, 12) * (3.0 / count)).astype(np.float32)
        kind = i % 3
        rows[i, 0] = kind
        if kind == 0:
            rows[i, 1:7] = [*pos, *col[:3]]
        elif kind == 1:
            rows[i, 1:13] = [*pos, *(pos + np.float32([0.15, 0.0, 0.05])), *col[:6]]
        else:
            rows[i, 1:22] = [*pos, 0.1, 0.0, 0.0, 0.0, 0.0, 0.1, *col]
    return rows


@pytest.mark.parametrize("count", [16, 17, 40])
def test_more_lights_than_the_argument_block_holds(lib, count):
    """The reference has no limit on the number of lights.  Up to 16 travel in the kernel argument block, more go
    through device memory: direct lighting with hard and soft shadows (queue passes), and with mirror reflection
    (the bounce paths), against the oracle."""
    g, sc, orc, sh, bvh = setup(lib, "cornell")
    rows = _many_lights(count)
    lights, n = sc.light_array(rows)
    W, H = 64, 48
    c = cam("A", (W, H))
    params = rt.default_params(ray_multiplier=2)
    for feats in (rt.features(shading=1, hard_shadow=1, soft_shadow=1, multiple_rays=1), rt.features(shading=1, soft_shadow=1, recursive=1)):
        img, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
        ref, ostats = orc.render(c, feats, params, W, H, lights=rows)
        assert np.abs(img - ref).max() <= 2e-5, (count, np.abs(img - ref).max())
        assert stats.primary_rays == ostats.primary_rays and stats.bounce_rays == ostats.bounce_rays
    # glossy reflection: the bounce calls as wavefront passes, their lighting through the same queue passes
    feats = rt.features(shading=1, soft_shadow=1, glossy=1)
    gparams = rt.default_params(glossy_filter_size=4, ray_depth=2)
    img, stats = lib.render(bvh, c, lights, n, feats, gparams, W, H)
    ref, ostats = orc.render(c, feats, gparams, W, H, lights=rows)
    err = np.abs(img - ref).max(axis=2)
    assert (err > 1e-4).mean() <= 0.01 and gc.psnr(img, ref) >= 45.0, (count, err.max())
    assert stats.primary_rays == ostats.primary_rays and stats.bounce_rays == ostats.bounce_rays


def test_row_tiles_compose_to_the_full_frame(lib):
    g, sc, orc, sh, bvh = setup(lib, "cornell_para")
    W, H = 80, 56
    feats = rt.features(shading=1, soft_shadow=1, multiple_rays=1)
    params = rt.default_params(ray_multiplier=2)
    lights, n = sc.light_array()
    c = cam("A", (W, H))
    full, _ = lib.render(bvh, c, lights, n, feats, params, W, H)
    parts = []
    for lo, hi in ((40, 56), (17, 40), (0, 17)):  # top slice first: Screen order
        part, _ = lib.render(bvh, c, lights, n, feats, params, W, H, row_begin=lo, row_end=hi)
        parts.append(part)
    assert np.array_equal(np.concatenate(parts, axis=0), full)
    # interleaved row blocks, one call per rank: rank r of 3 renders blocks r, r + 3, ... of 16 rows
    tiling = importlib.import_module("ray-tracer_amd.tiling")
    frame = np.zeros_like(full)
    for r in range(3):
        part, _ = lib.render_blocks(bvh, c, lights, n, feats, params, W, H, 16, r, 3)
        at = 0
        for b in tiling.blocks_of_rank(r, 3, H, 16):
            top, bottom, _, _ = tiling.block_rows(b, H, 16)
            frame[top:bottom] = part[at:at + bottom - top]
            at += 16
        assert at - 16 + (bottom - top) == part.shape[0]
    assert np.array_equal(frame, full)
    ref, _ = orc.render(c, feats, params, W, H)
    assert np.abs(full - ref).max() <= IMG_TOL


def test_config1_full_size(lib):
    """Config 1: Cornell 512x512, shading + mirror recursion depth 4 + BVH, 1 spp."""
    g, sc, orc, sh, bvh = setup(lib, "cornell")
    lights, n = sc.light_array()
    params = rt.default_params(count_tests=1)
    img, stats = lib.render(bvh, cam("A", (512, 512)), lights, n, rt.features(shading=1, recursive=1), params, 512, 512)
    ref, ostats = orc.render(cam("A", (512, 512)), rt.features(shading=1, recursive=1), params, 512, 512)
    assert np.abs(img - ref).max() <= IMG_TOL
    rowsum = img.astype(np.float64).sum(axis=1)
    assert np.abs(rowsum - g["c1_rowsum"]).max() <= 512 * IMG_TOL
    assert stats.primary_rays == ostats.primary_rays == 512 * 512
    assert stats.bounce_rays == ostats.bounce_rays
    # the kernels skip the direct lighting the reference computes and then discards on mirror hits, so they
    # run fewer triangle and box tests than the oracle counts, never more
    assert 0 < stats.box_tests <= ostats.box_tests
    assert 0 < stats.triangle_tests <= ostats.triangle_tests


def test_empty_scene_and_degenerate_sizes(lib):
    empty = rt.SceneData([])
    sh = lib.scene_create(empty)
    bvh = lib.bvh_build(sh)
    lights, n = empty.light_array()
    img, stats = lib.render(bvh, cam("A", (8, 8)), lights, n, rt.features(shading=1), rt.default_params(), 8, 8)
    assert not img.any() and stats.primary_rays == 64
    img, _ = lib.render(bvh, cam("A", (8, 8)), lights, n, rt.features(shading=1), rt.default_params(), 8, 8, row_begin=3, row_end=3)
    assert img.shape == (0, 8, 3)
    rays = np.zeros(0, rt.RAY_DTYPE)
    assert len(lib.intersect(bvh, rays, rt.features())) == 0


def test_bumpy_mesh_soft_shadows_match_oracle(lib):
    """The benchmark's workload class at a size the oracle finishes in seconds: displaced icosphere (5,120
    triangles, 12 levels), area light above it, 4 spp x 16 light samples."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    sc = rt.SceneData([synth.dragon_class_mesh(4)], lights=synth.area_light())
    orc = o.OracleScene(sc)
    bvh = lib.bvh_build(lib.scene_create(sc))
    W, H = 160, 96
    c = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0))
    feats = rt.features(shading=1, soft_shadow=1, multiple_rays=1)
    params = rt.default_params(ray_multiplier=2, count_tests=1)
    lights, n = sc.light_array()
    img, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    ref, ostats = orc.render(c, feats, params, W, H)
    assert np.abs(img - ref).max() <= IMG_TOL
    assert stats.primary_rays == ostats.primary_rays == W * H * 4
    # shadow queries whose Phong factor is exactly zero are answered without a ray
    assert stats.shadow_rays + stats.culled_shadow_rays == ostats.shadow_rays
    assert stats.culled_shadow_rays > 0 and stats.shadow_rays > 0


def test_bumpy_mesh_depth_of_field_soft_shadows_match_oracle(lib):
    """Depth of field through the queue passes: the centre rays' pass decides per pixel whether the 15 lens rays
    are traced (wherever the centre ray ends up being finished: in the budgeted kernel, the stream kernel or the
    unbudgeted one), the lens rays are cut into passes by the hit queue's capacity, soft shadows on every hit.
    Also as row blocks of two ranks."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    sc = rt.SceneData([synth.dragon_class_mesh(4)], lights=synth.area_light())
    orc = o.OracleScene(sc)
    bvh = lib.bvh_build(lib.scene_create(sc))
    W, H = 128, 80
    c = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0))
    feats = rt.features(shading=1, soft_shadow=1, depth_of_field=1)
    params = rt.default_params(dof_samples=15, dof_f=2.4, dof=0.02, dof_aperture=1.4, count_tests=1)
    lights, n = sc.light_array()
    img, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    ref, ostats = orc.render(c, feats, params, W, H)
    assert np.abs(img - ref).max() <= IMG_TOL
    # (the reference also probes every pixel with a copy of the centre ray, render.cpp:106-109: the oracle counts
    # those W * H queries as probe rays; here the centre ray's own traversal answers the probe)
    assert stats.primary_rays == ostats.primary_rays
    assert W * H < stats.primary_rays < W * H * 16  # some pixels got their lens rays, not all
    assert stats.shadow_rays + stats.culled_shadow_rays == ostats.shadow_rays
    assert stats.culled_shadow_rays > 0
    # two ranks' interleaved row blocks give the same frame
    parts = []
    for rank in range(2):
        part, _ = lib.render_blocks(bvh, c, lights, n, feats, params, W, H, 16, rank, 2)
        parts.append(part)
    whole = np.empty_like(img)
    for rank in range(2):
        rows = [r for b in range(rank, (H + 15) // 16, 2) for r in range(b * 16, min(b * 16 + 16, H))]
        whole[rows] = parts[rank].reshape(-1, W, 3)
    assert np.array_equal(whole, img)


@pytest.mark.skipif(os.environ.get("RT_TEST_CHILD") == "1", reason="already the child run")
@pytest.mark.parametrize("knobs", [dict(RT_PAIR_STACK_LDS="64", RT_PRIMARY_BUDGET="2"), dict(RT_PRIMARY_STACK="1"), dict(RT_PRIMARY_STACK="3")],
                         ids=["budget-2", "stack-1", "stack-3"])
def test_small_limits_take_the_spill_and_give_up_paths(knobs):
    """Same image tests in a child process with the pair stack's LDS share at its minimum (every deeper stack spills
    to device memory) and a camera-ray budget of two nodes (nearly every ray that enters the root box goes through
    the give-up queue, the stream kernel and the unbudgeted traversal); and with a camera-ray stack of one or three
    entries (rays give up in the middle of their traversal, many of them with a hit already accepted)."""
    env = dict(os.environ, RT_TEST_CHILD="1", **knobs)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.abspath(__file__),
           "-k", "bumpy or row_tiles or (images_match and (soft or shading or hard or sky))"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


@pytest.mark.skipif(os.environ.get("RT_TEST_CHILD") == "1", reason="already the child run")
@pytest.mark.parametrize("mode", ["0", "1"])
def test_bounce_frames_in_both_evaluation_orders(mode):
    """Frames with mirror reflection, see-through continuation, glossy reflection or the reflection map can be evaluated
    by the per-lane kernel (one lane walks a pixel sample's whole call tree) or as wavefront passes (one generation of
    calls per pass).  By default glossy frames take the wavefront passes and the others the per-lane kernel; here every
    image test runs again in a child process with all bounce frames forced one way (RT_WAVEFRONT=0 / 1)."""
    env = dict(os.environ, RT_TEST_CHILD="1", RT_WAVEFRONT=mode)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.abspath(__file__),
           "-k", "images_match or glossy or config or wavefront_nodes"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


@pytest.mark.skipif(os.environ.get("RT_TEST_CHILD") == "1", reason="already the child run")
def test_wavefront_chunks_that_overflow_are_split():
    """A node buffer of 6,000 nodes (RT_WAVEFRONT_NODES) cannot hold the call trees of a chunk of pixel samples once glossy
    calls fan out 64-fold: chunks are abandoned and rendered again in halves until they fit.  The images (and the ray
    counts, which must not keep anything of an abandoned attempt) are the same."""
    env = dict(os.environ, RT_TEST_CHILD="1", RT_WAVEFRONT="1", RT_WAVEFRONT_NODES="6000")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.abspath(__file__),
           "-k", "glossy or config_feature_sets or config1 or (images_match and (recursive or transparency or sky_refl))"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_wavefront_nodes_are_reported(lib):
    """rt_stats.wavefront_nodes says which way a frame went.  Left to itself the library sends every glossy frame through
    the wavefront passes; a direct-lighting frame never takes them, a mirror frame only when forced."""
    g, sc, orc, sh, bvh = setup(lib, "transparent_cornell")
    lights, n = sc.light_array()
    W, H = 64, 40
    forced = os.environ.get("RT_WAVEFRONT")
    _, st = lib.render(bvh, cam("A", (W, H)), lights, n, rt.features(shading=1, glossy=1, transparency=1, hard_shadow=1), rt.default_params(glossy_filter_size=16), W, H)
    if forced == "0":
        assert st.wavefront_nodes == 0
    else:
        assert st.wavefront_nodes >= W * H and st.bounce_rays > 0
    # no shadow rays and a fifth of the area specular: passes all the same
    _, st = lib.render(bvh, cam("A", (W, H)), lights, n, rt.features(shading=1, glossy=1, transparency=1), rt.default_params(glossy_filter_size=16), W, H)
    assert (st.wavefront_nodes > 0) == (forced != "0")
    # everything specular (the monkey): passes unless switched off
    g2, sc2, orc2, sh2, bvh2 = setup(lib, "monkey")
    lights2, n2 = sc2.light_array()
    _, st = lib.render(bvh2, cam("A", (W, H)), lights2, n2, rt.features(shading=1, glossy=1), rt.default_params(glossy_filter_size=4, ray_depth=2), W, H)
    assert (st.wavefront_nodes > 0) == (forced != "0")
    _, st = lib.render(bvh, cam("A", (W, H)), lights, n, rt.features(shading=1, hard_shadow=1), rt.default_params(), W, H)
    assert st.wavefront_nodes == 0
    # mirror recursion: wavefront only when forced
    _, st = lib.render(bvh, cam("A", (W, H)), lights, n, rt.features(shading=1, recursive=1), rt.default_params(), W, H)
    assert (st.wavefront_nodes > 0) == (forced == "1")


def test_headline_frame_at_full_size(lib):
    """The benchmark's frame itself (1920x1080, 16 spp, 16 light samples, 81,920 triangles): bands of it against the
    oracle, plus the size-independent properties — two renders are identical, and the interleaved shares of 8 ranks
    compose to the same frame bit for bit."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    tiling = importlib.import_module("ray-tracer_amd.tiling")
    sc = rt.SceneData([synth.dragon_class_mesh(6)], lights=synth.area_light())
    orc = o.OracleScene(sc)
    bvh = lib.bvh_build(lib.scene_create(sc))
    W, H = 1920, 1080
    c = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0))
    feats = rt.features(shading=1, soft_shadow=1, multiple_rays=1)
    params = rt.default_params(ray_multiplier=4)
    lights, n = sc.light_array()
    full, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    assert stats.primary_rays == W * H * 16
    assert (stats.shadow_rays + stats.culled_shadow_rays) % 16 == 0  # 16 light samples per hit, traced or culled
    for lo, hi in ((539, 542), (250, 251), (905, 906)):  # rows counted from the bottom, like rt_render
        ref, _ = orc.render(c, feats, params, W, H, row_begin=lo, row_end=hi)
        assert np.abs(full[H - hi:H - lo] - ref).max() <= IMG_TOL
    again, _ = lib.render(bvh, c, lights, n, feats, params, W, H)
    assert np.array_equal(again, full)
    frame = np.zeros_like(full)
    for r in range(8):
        part, _ = lib.render_blocks(bvh, c, lights, n, feats, params, W, H, tiling.ROW_BLOCK, r, 8)
        for k, b in enumerate(tiling.blocks_of_rank(r, 8, H)):
            top, bottom, _, _ = tiling.block_rows(b, H)
            frame[top:bottom] = part[k * tiling.ROW_BLOCK:k * tiling.ROW_BLOCK + bottom - top]
    assert np.array_equal(frame, full)


def _compose_shares(lib, bvh, c, lights, n, feats, params, W, H, world):
    """The frame put together from the interleaved row-block shares of `world` ranks."""
    tiling = importlib.import_module("ray-tracer_amd.tiling")
    frame = np.zeros((H, W, 3), np.float32)
    for r in range(world):
        part, _ = lib.render_blocks(bvh, c, lights, n, feats, params, W, H, tiling.ROW_BLOCK, r, world)
        for k, b in enumerate(tiling.blocks_of_rank(r, world, H)):
            top, bottom, _, _ = tiling.block_rows(b, H)
            frame[top:bottom] = part[k * tiling.ROW_BLOCK:k * tiling.ROW_BLOCK + bottom - top]
    return frame


@pytest.mark.parametrize("scene", ["monkey", "cube_textured"])
def test_config2_at_full_size(lib, scene):
    """Config 2 (1920x1080, textures + sky box + 4 spp; nearest texel as in the reference, and the bilinear lookup this
    library defines): row bands against the oracle, determinism, composition from 4 ranks' shares."""
    g, sc, orc, sh, bvh = setup(lib, scene)
    W, H = 1920, 1080
    c = cam("A", (W, H))
    lights, n = sc.light_array()
    params = rt.default_params(ray_multiplier=2, enabled_skybox=1)
    for bilinear in (0, 1):
        feats = rt.features(shading=1, texture_mapping=1, environment_mapping=1, multiple_rays=1, bilinear_filtering=bilinear)
        full, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
        assert stats.primary_rays == W * H * 4
        for lo, hi in ((538, 542), (300, 302), (800, 801)):
            ref, _ = orc.render(c, feats, params, W, H, row_begin=lo, row_end=hi)
            assert np.abs(full[H - hi:H - lo] - ref).max() <= IMG_TOL
        again, _ = lib.render(bvh, c, lights, n, feats, params, W, H)
        assert np.array_equal(again, full)
        assert np.array_equal(_compose_shares(lib, bvh, c, lights, n, feats, params, W, H, 4), full)


@pytest.mark.parametrize("scene", gc.C4_SCENES)
def test_config4_at_full_size(lib, scene):
    """Config 4 (1920x1080, transparency + glossy reflection with 64 samples per bounce, 36 spp): row bands against the
    oracle to the glossy bar and PSNR >= 45 dB, no transparency cap hit, determinism, composition from 4 shares."""
    g, sc, orc, sh, bvh = setup(lib, scene)
    W, H = 1920, 1080
    c = cam("A", (W, H))
    lights, n = sc.light_array()
    feats = rt.features(shading=1, transparency=1, glossy=1, multiple_rays=1)
    params = rt.default_params(ray_multiplier=6, glossy_filter_size=64)
    full, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    assert stats.primary_rays == W * H * 36
    assert stats.transparency_overflows == 0
    assert stats.bounce_rays > 10 * W * H  # glossy fan-out and see-through continuations
    for lo, hi in ((540, 542), (420, 421), (700, 701)):
        ref, _ = orc.render(c, feats, params, W, H, row_begin=lo, row_end=hi)
        band = full[H - hi:H - lo]
        err = np.abs(band - ref).max(axis=2)
        assert (err > 1e-4).mean() <= 0.01, f"rows {lo}-{hi}: {(err > 1e-4).sum()} pixels differ, max {err.max()}"
        assert gc.psnr(band, ref) >= 45.0
    again, _ = lib.render(bvh, c, lights, n, feats, params, W, H)
    assert np.array_equal(again, full)
    assert np.array_equal(_compose_shares(lib, bvh, c, lights, n, feats, params, W, H, 4), full)


def test_config5_soft_shadows_depth_of_field_bloom(lib):
    """Config 5's feature set in the form the reference can express (depth of field and supersampling exclude each
    other there): dragon-class mesh, 16-sample soft shadows, 63 lens rays per pixel, bloom with the reference's
    stride-by-height taps on a non-square frame.  Reduced size against the oracle; 3840x2160 through the
    size-independent properties: determinism, and the frame composed from 8 ranks' shares then bloomed on one rank
    equals the whole frame bloomed."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    sc = rt.SceneData([synth.dragon_class_mesh(6)], lights=synth.area_light())
    orc = o.OracleScene(sc)
    bvh = lib.bvh_build(lib.scene_create(sc))
    feats = rt.features(shading=1, soft_shadow=1, depth_of_field=1, bloom=1)
    params = rt.default_params(dof_samples=63, dof_f=2.5, dof=0.05, dof_aperture=1.4, bloom_threshold=0.3)
    lights, n = sc.light_array()
    W, H = 112, 63
    c = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0))
    img, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    ref, ostats = orc.render(c, feats, params, W, H)
    assert np.abs(img - ref).max() <= IMG_TOL
    assert stats.primary_rays == ostats.primary_rays and stats.primary_rays > W * H * 8
    bloomed = lib.bloom(img.copy(), params)
    assert np.abs(bloomed - o.bloom(ref, params)).max() <= IMG_TOL
    assert not np.array_equal(bloomed, img)  # the threshold is low enough for the filter to add light
    W, H = 3840, 2160
    c = rt.make_camera(width=W, height=H, fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0))
    full, stats = lib.render(bvh, c, lights, n, feats, params, W, H)
    assert W * H < stats.primary_rays <= W * H * 64
    again, _ = lib.render(bvh, c, lights, n, feats, params, W, H)
    assert np.array_equal(again, full)
    composed = _compose_shares(lib, bvh, c, lights, n, feats, params, W, H, 8)
    assert np.array_equal(composed, full)
    assert np.array_equal(lib.bloom(composed, params), lib.bloom(full.copy(), params))
