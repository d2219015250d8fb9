"""The C++ host facade (ray-tracer_amd/host): the reference's own surface — Scene, BvhInterface, Trackball,
Screen, renderRayTracing — over the C ABI.  The driver facade_render mirrors the reference's CLI flow."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as o

rt = o.rt
HOST = os.path.join(o.ROOT, "ray-tracer_amd", "host")
DRIVER = os.path.join(HOST, "facade_render")


def run_driver(tmp_path, scene_name, flags, size=o.SIZE_A, bmp=False):
    g = o.golden(scene_name)
    sc = rt.SceneData.from_golden(g)
    scene_file = str(tmp_path / "scene.rtscene")
    sc.write_binary(scene_file)
    out = str(tmp_path / "frame.f32")
    cmd = [DRIVER, scene_file, out, str(size[0]), str(size[1]), flags]
    if bmp:
        cmd.append(str(tmp_path / "frame.bmp"))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    return g, res, out


def test_facade_builds_reference_hierarchy_without_gpu(tmp_path):
    assert os.path.exists(DRIVER), "build it with __graft_entry__.build()"
    g, res, out = run_driver(tmp_path, "cornell", "shading")
    assert res.returncode == 0, res.stderr
    levels, leaves = g["bvh.levels_leaves"]
    assert f"levels {levels} leaves {leaves}" in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("scene,flags,tag", [
    ("cornell", "shading,recursive", "A.recursive"),
    ("cornell", "shading,recursive,bloom", "A.bloom_gauss"),
    ("cube_textured", "shading,recursive,hard,normals,textures", "A.all_det"),
    ("transparent_cubes", "shading,hard,transparency", "A.transparency"),
    ("cornell_para", "shading,soft,multi2", "A.multi2_soft"),
    ("spheres", "shading,hard", "A.hard"),
])
def test_facade_render_matches_reference(tmp_path, scene, flags, tag):
    g, res, out = run_driver(tmp_path, scene, flags, bmp=True)
    assert res.returncode == 0 and "renderRayTracing:" not in res.stderr, res.stderr
    W, H = o.SIZE_A
    img = np.fromfile(out, dtype=np.float32).reshape(H, W, 3)
    ref = g["img." + tag]
    assert np.abs(img - ref).max() <= 2e-6
    # the centre debug ray goes through BvhInterface::intersect
    rays = o.camera_rays(rt.make_camera(width=W, height=H, **o.CAM_A), 2, 2)  # ndc (0,0) is pixel (1,1) of a 2x2 frame
    hit = o.OracleScene(rt.SceneData.from_golden(g)).intersect(rays[3:4], rt.features())
    assert f"centre ray hit {int(hit['hit'][0])}" in res.stdout
    # BMP: 54-byte header + 32-bit pixels, bottom row first, channel = trunc(clamp(c) * 255)
    bmp = np.fromfile(str(tmp_path / "frame.bmp"), dtype=np.uint8)
    assert bmp[0] == ord("B") and bmp[1] == ord("M") and len(bmp) == 54 + W * H * 4
    px = bmp[54:].reshape(H, W, 4)[::-1, :, 2::-1]
    want = (np.clip(img, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(px, want)
