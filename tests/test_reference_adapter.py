"""The reference-side adapter (ray-tracer_amd/host/reference_adapter): the two translation units a maintainer of the
reference swaps in, compiled against the reference's OWN headers (src/bvh_interface.h, render.h, scene.h, common.h,
screen.h, framework/trackball.h, vendored glm) and linked with the reference's remaining objects and librt_amd.so by
`make -C oracle adapter`.  The build needs /root/reference (the build container); the program it produces,
oracle/_ref/adapter_demo, travels to the GPU box and is run there on scenes written from the golden fixtures."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as o

rt = o.rt
DEMO = os.path.join(o.ROOT, "oracle", "_ref", "adapter_demo")
HAVE_REFERENCE = os.path.exists("/root/reference/src/bvh_interface.h")


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference checkout (build container only)")
def test_adapter_compiles_against_the_reference_headers_and_links(tmp_path):
    res = subprocess.run(["make", "-s", "-C", os.path.join(o.ROOT, "oracle"), "adapter"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "warning" not in (res.stdout + res.stderr)  # the two adapter files are compiled with -Wall -Wextra
    assert os.path.exists(DEMO)
    # no GPU here: the hierarchy is still built (on the host, in the reference's node order) through the adapter
    g = o.golden("cornell")
    scene_file = str(tmp_path / "scene.rtscene")
    rt.SceneData.from_golden(g).write_binary(scene_file)
    run = subprocess.run([DEMO, scene_file, str(tmp_path / "gpu.f32"), str(tmp_path / "cpu.f32"), "8", "8", "shading"], capture_output=True, text=True, timeout=120)
    levels, leaves = g["bvh.levels_leaves"]
    assert f"levels {levels} leaves {leaves}" in run.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/adapter_demo is built where the reference checkout exists")
@pytest.mark.parametrize("scene,flags,tag", [
    ("cornell", "shading,recursive", "A.recursive"),
    ("cornell", "shading,recursive,bloom", "A.bloom_gauss"),
    ("cube_textured", "shading,recursive,hard,normals,textures", "A.all_det"),
    ("transparent_cubes", "shading,hard,transparency", "A.transparency"),
    ("spheres", "shading,hard", "A.hard"),
])
def test_swapped_in_adapter_renders_the_reference_frame(tmp_path, scene, flags, tag):
    """Both render functions of the linked program against the compiled reference's golden frame: the adapter's
    renderRayTracing (one rt_render call), and the reference's own pixel loop / getFinalColor / light.cpp on top of the
    adapter's BvhInterface::intersect (one rt_intersect_batch call per ray)."""
    g = o.golden(scene)
    scene_file = str(tmp_path / "scene.rtscene")
    rt.SceneData.from_golden(g).write_binary(scene_file)
    W, H = o.SIZE_A
    gpu, cpu = str(tmp_path / "gpu.f32"), str(tmp_path / "cpu.f32")
    run = subprocess.run([DEMO, scene_file, gpu, cpu, str(W), str(H), flags], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "renderRayTracing" not in run.stderr and "BvhInterface" not in run.stderr, run.stderr[-2000:]
    ref = g["img." + tag]
    a = np.fromfile(gpu, dtype=np.float32).reshape(H, W, 3)
    b = np.fromfile(cpu, dtype=np.float32).reshape(H, W, 3)
    assert np.abs(a - ref).max() <= 2e-6  # device powf against glibc pow in the Phong term
    assert np.array_equal(b, ref)         # host arithmetic of the reference itself; only the hit records come from the GPU
