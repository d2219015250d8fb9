"""CPU tests: the oracle (oracle/cpu_ref.cpp) against the golden vectors produced from the compiled
reference (tests/golden/, see oracle/gen_golden.py).  Bit-exact unless a tolerance is stated."""
import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as o

rt = o.rt
_cache = {}


def scene_and_oracle(name, sah=False):
    key = (name, sah)
    if key not in _cache:
        g = o.golden(name)
        sc = rt.SceneData.from_golden(g)
        _cache[key] = (g, sc, o.OracleScene(sc, sah=sah))
    return _cache[key]


def cam(which, size):
    spec = o.CAM_A if which == "A" else o.CAM_B
    return rt.make_camera(width=size[0], height=size[1], **spec)


def size_of(which):
    return o.SIZE_A if which == "A" else o.SIZE_B


@pytest.mark.parametrize("scene", gc.SCENES)
def test_bvh_node_order(scene):
    g, sc, orc = scene_and_oracle(scene)
    assert list(orc.levels_leaves()) == list(g["bvh.levels_leaves"])
    meta, box, ids = orc.export()
    assert np.array_equal(meta, g["bvh.node_meta"])
    assert np.array_equal(box, g["bvh.node_box"])
    assert np.array_equal(ids, g["bvh.node_ids"])


@pytest.mark.parametrize("scene", ["cornell", "monkey"])
def test_bvh_sah_node_order(scene):
    g, sc, orc = scene_and_oracle(scene, sah=True)
    assert list(orc.levels_leaves()) == list(g["bvh_sah.levels_leaves"])
    meta, box, ids = orc.export()
    assert np.array_equal(meta, g["bvh_sah.node_meta"])
    assert np.array_equal(box, g["bvh_sah.node_box"])
    assert np.array_equal(ids, g["bvh_sah.node_ids"])


@pytest.mark.parametrize("which", ["A", "B"])
def test_camera_rays(which):
    g = o.golden("cornell")
    W, H = size_of(which)
    rays = o.camera_rays(cam(which, (W, H)), W, H)
    ref = o.rays_from_array(g[f"{which}.prune.rays"])
    for k in ("origin", "direction", "t"):
        assert np.array_equal(rays[k], ref[k]), k


def _hit_cases():
    out = []
    for s in gc.SCENES:
        out.append((s, "A", "prune"))
        out.append((s, "A", "noprune"))
        if s != "teapot":
            out.append((s, "A", "brute"))
        if s in gc.B_SCENES:
            out.append((s, "B", "prune"))
            out.append((s, "B", "noprune"))
    return out


@pytest.mark.parametrize("scene,which,mode", _hit_cases())
def test_hit_records(scene, which, mode):
    g, sc, orc = scene_and_oracle(scene)
    W, H = size_of(which)
    rays = o.camera_rays(cam(which, (W, H)), W, H)
    feats = {"prune": rt.features(normal_interp=1, texture_mapping=1), "noprune": rt.features(recursive=1),
             "brute": rt.features(normal_interp=1, texture_mapping=1, accel_structure=0)}[mode]
    h = orc.intersect(rays, feats)
    p = f"{which}.{mode}."
    assert np.array_equal(h["hit"], g[p + "hit"])
    assert np.array_equal(h["t"], g[p + "t"])
    m = h["hit"] == 1
    if mode != "brute":  # the reference only exposes the final triangle on the BVH path
        assert np.array_equal(h["mesh"][m], g[p + "mesh"][m])
        tri_known = m & (g[p + "mesh"] >= 0)  # the reference does not expose which sphere was hit
        assert np.array_equal(h["triangle"][tri_known], g[p + "tri"][tri_known])
    if p + "normal" in g:
        assert np.array_equal(h["normal"][m], g[p + "normal"][m])
        assert np.array_equal(h["bary"][m], g[p + "bary"][m])
        assert np.array_equal(h["kd"][m], g[p + "kd"][m])


@pytest.mark.parametrize("scene", gc.SCENES)
@pytest.mark.parametrize("mode", ["prune", "noprune"])
def test_shadow_visibility(scene, mode):
    g, sc, orc = scene_and_oracle(scene)
    W, H = o.SIZE_A
    rays = o.camera_rays(cam("A", (W, H)), W, H)
    feats = rt.features(shading=1, hard_shadow=1, soft_shadow=1, recursive=int(mode == "noprune"))
    h = orc.intersect(rays, feats)
    m = h["hit"] == 1
    hit_rays = rays[m].copy()
    hit_rays["t"] = h["t"][m]
    vis = orc.visibility(hit_rays, g[f"A.{mode}.vis_samples"], feats)
    assert np.array_equal(vis, g[f"A.{mode}.vis"][m])


def _image_cases():
    return [(s, t) for s in gc.SCENES for t in gc.image_tags(s)]


@pytest.mark.parametrize("scene,tag", _image_cases())
def test_images(scene, tag):
    g, sc, orc = scene_and_oracle(scene)
    which, feats, params = gc.IMAGE_CASES[tag]
    W, H = size_of(which)
    img, stats = orc.render(cam(which, (W, H)), feats, params, W, H)
    if feats.bloom:
        img = o.bloom(img, params)
    ref = g["img." + tag]
    assert img.shape == ref.shape
    assert np.array_equal(img, ref), f"max abs diff {np.abs(img - ref).max()} on {(img != ref).any(axis=2).sum()} pixels"


@pytest.mark.parametrize("scene", gc.GLOSSY_SCENES)
def test_glossy_images(scene):
    g, sc, orc = scene_and_oracle(scene)
    for tag, (which, (W, H), feats, params) in gc.GLOSSY_CASES.items():
        if "img." + tag not in g:
            continue
        img, _ = orc.render(cam(which, (W, H)), feats, params, W, H)
        ref = g["img." + tag]
        # The Gaussian deviates go through double log/cos/sqrt; identical libm here, so exact.
        assert np.array_equal(img, ref), f"{tag}: max abs diff {np.abs(img - ref).max()}"


def _config_cases():
    return [(s, t) for s in gc.C2_SCENES for t in ("A.c2", "B.c2")] + [(s, "A.c4") for s in gc.C4_SCENES]


@pytest.mark.parametrize("scene,tag", _config_cases())
def test_config_feature_sets_at_fixture_size(scene, tag):
    """Configs 2 and 4 (sky box + textures + 4 spp; transparency + glossy with 64 samples per bounce + 4 spp) against
    the compiled reference's frames: bit-identical."""
    g, sc, orc = scene_and_oracle(scene)
    which, size, feats, params = gc.CONFIG_CASES[tag]
    W, H = size or size_of(which)
    img, _ = orc.render(cam(which, (W, H)), feats, params, W, H)
    ref = o.golden("configs")[f"{scene}.img.{tag}"]
    assert np.array_equal(img, ref), f"max abs diff {np.abs(img - ref).max()}"


def test_config1_full_size_digest():
    """Config 1 (Cornell 512x512, shading + mirror recursion depth 4 + BVH, 1 spp): bit-identical frame."""
    g, sc, orc = scene_and_oracle("cornell")
    img, stats = orc.render(cam("A", (512, 512)), rt.features(shading=1, recursive=1), rt.default_params(), 512, 512)
    rowsum = img.astype(np.float64).sum(axis=1)
    assert np.array_equal(rowsum, g["c1_rowsum"])
    h = 1469598103934665603
    data = img.tobytes()
    # FNV-1a 64 in numpy-free python would be slow over 3 MB; do it vectorised per byte is impossible
    # (sequential dependency), so check it on the raw bytes with a C-speed loop via int.from_bytes chunks.
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert [h & 0xFFFFFFFF, h >> 32] == [int(v) for v in g["c1_fnv1a64"]]
    assert stats.primary_rays == 512 * 512


KATS = {"aabb": 2, "tri": 8, "shade": 10, "env": 3, "interp": 5}


@pytest.mark.parametrize("name", list(KATS))
def test_known_answers(name):
    k = o.golden("kats")
    out = o.kat(name, k[name + "_in"], KATS[name])
    ref = k[name + "_out"]
    same = (out == ref) | (np.isnan(out) & np.isnan(ref))
    assert same.all(), f"{(~same).any(axis=1).sum()} of {len(ref)} rows differ"


def test_texture_fetch_known_answers():
    import ctypes as C
    k = o.golden("kats")
    g, sc, orc = scene_and_oracle("cube_textured")
    uv = np.ascontiguousarray(k["texel_in"])
    out = np.zeros((len(uv), 3), np.float32)
    o.lib().orc_kat_texel(orc.h, 0, uv.ctypes.data, len(uv), out.ctypes.data)
    # u = 0 (and any u*W < 0.5) makes the reference index texel -1: it reads outside the image (undefined
    # behaviour, texture.cpp:14-17), so those rows of the fixture hold garbage and are excluded; the
    # oracle and the kernels clamp the index instead.
    tex = sc.textures[0]
    th, tw = tex.shape[0], tex.shape[1]
    def c_round(x):  # std::round: halves away from zero (np.round goes to even)
        return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))
    i = np.fmod(c_round((uv[:, 0] * np.float32(tw)).astype(np.float64) - 0.5), tw).astype(np.float32)
    j = np.fmod(c_round(((np.float32(1) - uv[:, 1]) * np.float32(th)).astype(np.float64) - 0.5), th).astype(np.float32)
    idx = (j * np.float32(tw) + i).astype(np.int64)
    ok = (idx >= 0) & (idx < tw * th)
    assert ok.sum() > 200 and (~ok).sum() > 0
    assert np.array_equal(out[ok], k["texel_out"][ok])
    flat = tex.reshape(-1, 3).astype(np.float32) / np.float32(255.0)
    assert np.array_equal(out[~ok], flat[np.clip(idx[~ok], 0, tw * th - 1)])
    uvf = np.ascontiguousarray(k["clamp_in"])
    out = np.zeros((len(uvf), 3), np.float32)
    o.lib().orc_kat_clamp(orc.h, uvf.ctypes.data, len(uvf), out.ctypes.data)
    assert np.array_equal(out, k["clamp_out"])


def test_bilinear_texture_lookup_definition():
    """Bilinear filtering has no reference behaviour (the reference's acquireTexel ignores its switch): PARITY UNPINNED.
    The oracle's definition is checked against an independent numpy statement of it and against the nearest lookup
    at the texel centres, where the two must agree."""
    g, sc, orc = scene_and_oracle("cube_textured")
    tex = sc.textures[0]
    th, tw = tex.shape[0], tex.shape[1]
    flat = tex.reshape(-1, 3).astype(np.float32) / np.float32(255.0)
    rng = np.random.default_rng(11)
    uv = rng.random((4000, 2), dtype=np.float32) * np.float32(1.5) - np.float32(0.25)  # also outside [0, 1): wraps
    uv[:8] = [[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5], [0.999999, 1e-7], [-0.25, 1.25], [1.0 / tw, 1.0 / th]]
    out = np.zeros((len(uv), 3), np.float32)
    o.lib().orc_kat_texel_bilinear(orc.h, 0, uv.ctypes.data, len(uv), out.ctypes.data)
    f32 = np.float32
    x = uv[:, 0] * f32(tw) - f32(0.5)
    y = (f32(1) - uv[:, 1]) * f32(th) - f32(0.5)
    x0, y0 = np.floor(x), np.floor(y)
    fx, fy = (x - x0).astype(f32), (y - y0).astype(f32)
    def wrap(c, n):
        return np.minimum(np.mod(c.astype(np.int64), n), n - 1)
    i0, i1, j0, j1 = wrap(x0, tw), wrap(x0 + 1, tw), wrap(y0, th), wrap(y0 + 1, th)
    t = lambda j, i: flat[j * tw + i]
    top = t(j0, i0) * (f32(1) - fx)[:, None] + t(j0, i1) * fx[:, None]
    bottom = t(j1, i0) * (f32(1) - fx)[:, None] + t(j1, i1) * fx[:, None]
    want = top * (f32(1) - fy)[:, None] + bottom * fy[:, None]
    assert np.array_equal(out, want.astype(f32))
    # texel centres: weights are exactly (1, 0), so the result is the nearest lookup's texel
    ci, cj = np.meshgrid(np.arange(tw), np.arange(th))
    centres = np.stack([(ci.ravel() + 0.5) / tw, 1.0 - (cj.ravel() + 0.5) / th], axis=1).astype(f32)
    a = np.zeros((len(centres), 3), f32)
    b = np.zeros((len(centres), 3), f32)
    o.lib().orc_kat_texel_bilinear(orc.h, 0, centres.ctypes.data, len(centres), a.ctypes.data)
    o.lib().orc_kat_texel(orc.h, 0, centres.ctypes.data, len(centres), b.ctypes.data)
    assert np.array_equal(a, b)
