"""The facade's image decoder (ray-tracer_amd/host/image_loader.cpp) and scene table (scene_table.cpp) against the
reference (Image(path) on stb_image 2.27, loadScenePrebuilt):
  - on the PNG and JPEG files under tests/golden/images/, which were made for this repository
    (oracle/gen_image_golden.py); expected bytes from the compiled reference, stored in tests/golden/images.npz;
    a file the reference refuses must be refused here too;
  - on the reference's own image and scene files, when its checkout is present on this machine.
Bit-exact: every byte of every pixel; every light, sphere, vertex and triangle of the scene table."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as o
from test_obj_loader import REFERENCE_DATA, load_with_facade, read_meshes, same_bits

DRIVER = os.path.join(o.ROOT, "ray-tracer_amd", "host", "facade_render")
IMAGES = os.path.join(o.ROOT, "tests", "golden", "images")
REF_DUMP = os.path.join(o.ROOT, "oracle", "_ref", "ref_dump")
SCENE_TYPES = {"single_triangle": 0, "cube_segment": 1, "cube_textured": 2, "cornell": 3, "cornell_para": 4, "monkey": 5, "teapot": 6, "spheres": 8, "quad_textured": 10,
               "transparent_cubes": 11, "transparent_cornell": 12}
SCENE_FILES = {0: "triangle.obj", 1: "cube.obj", 2: "cube-textured.obj", 3: "CornellBox-Mirror-Rotated.obj", 4: "CornellBox-Mirror-Rotated.obj", 5: "monkey.obj", 6: "teapot.obj",
               7: "dragon.obj", 9: "custom.obj", 10: "quadtextured.obj", 11: "transparent-cubes.obj", 12: "cornelltransparent.obj"}


def decode(tmp_path, image_file):
    """(exit status, pixels or None)"""
    out = str(tmp_path / "image.bin")
    res = subprocess.run([DRIVER, "--image", image_file, out], capture_output=True, text=True, timeout=120)
    if res.returncode != 0:
        return res.returncode, None
    b = open(out, "rb").read()
    w, h = np.frombuffer(b[:8], "<i4")
    return 0, np.frombuffer(b[8:], np.uint8).reshape(h, w, 3)


def test_fixture_set_is_complete():
    g = np.load(os.path.join(o.ROOT, "tests", "golden", "images.npz"))
    assert sorted(g.keys()) == sorted(os.listdir(IMAGES))
    kinds = {k.split(".")[-1] for k in g.keys()}
    assert kinds == {"png", "jpg"}
    assert sum(1 for k in g.keys() if g[k].size == 0) >= 8  # the refused ones


@pytest.mark.parametrize("kind", ["png", "jpg"])
def test_own_image_files_decode_like_the_reference(tmp_path, kind):
    g = np.load(os.path.join(o.ROOT, "tests", "golden", "images.npz"))
    names = sorted(k for k in g.keys() if k.endswith(kind))
    assert len(names) > 30
    for name in names:
        rc, px = decode(tmp_path, os.path.join(IMAGES, name))
        want = g[name]
        if want.size == 0:
            assert rc == 3, f"{name}: the reference refuses this file"
        else:
            assert rc == 0, name
            assert px.shape == want.shape, name
            assert np.array_equal(px, want), f"{name}: {(px != want).sum()} bytes differ"


def test_missing_file_is_refused(tmp_path):
    rc, _ = decode(tmp_path, str(tmp_path / "nothing_here.png"))
    assert rc == 3


def test_loader_attaches_texture_of_own_obj_file(tmp_path):
    """loadMesh decodes map_Kd files next to the OBJ file (reference framework/src/mesh.cpp:131-133); a missing one
    makes the whole load fail like the reference's does."""
    own = os.path.join(o.ROOT, "tests", "golden", "obj_own")
    work = tmp_path / "scene"
    shutil.copytree(own, work)
    _, _, table, _ = load_with_facade(tmp_path, str(work / "shapes.obj"), 0)
    assert table[:, 4].sum() >= 1
    os.remove(work / "checker.png")
    res = subprocess.run([DRIVER, "--obj", str(work / "shapes.obj"), "0", str(tmp_path / "x.rtscene")], capture_output=True, text=True, timeout=60)
    assert res.returncode != 0


def run_table(tmp_path, scene_type, data_dir, skybox=0):
    out = str(tmp_path / "table.rtscene")
    res = subprocess.run([DRIVER, "--table", str(scene_type), str(data_dir), str(skybox), out], capture_output=True, text=True, timeout=300)
    lights = [np.array(line.split()[1:], dtype=np.float32) for line in res.stdout.splitlines() if line.startswith("light ")]
    spheres = [np.array(line.split()[1:], dtype=np.float32) for line in res.stdout.splitlines() if line.startswith("sphere ")]
    sizes = [tuple(int(v) for v in line.split()[1:]) for line in res.stdout.splitlines() if line.startswith(("texture ", "skybox "))]
    return res.returncode, np.array(lights, np.float32).reshape(-1, 22), np.array(spheres, np.float32).reshape(-1, 12), sizes


@pytest.mark.parametrize("name", sorted(SCENE_TYPES))
def test_scene_table_lights_and_spheres(tmp_path, name):
    """Lights and spheres of every prebuilt scene against the arrays the reference's loadScenePrebuilt produced
    (reference src/scene.cpp:5-107).  The OBJ file the entry names is one of this repository's own, put under the
    expected name in a scratch data directory: the geometry is not what is checked here."""
    g = np.load(os.path.join(o.ROOT, "tests", "golden", name + ".npz"))
    t = SCENE_TYPES[name]
    data = tmp_path / "data"
    shutil.copytree(os.path.join(o.ROOT, "tests", "golden", "obj_own"), data)
    if t in SCENE_FILES:
        shutil.copy(data / "objects.obj", data / SCENE_FILES[t])
    rc, lights, spheres, _ = run_table(tmp_path, t, data)
    assert rc == 0
    assert same_bits(lights, g["scene.lights"])
    assert same_bits(spheres, g["scene.spheres"].reshape(-1, 12))


def test_scene_table_needs_its_files(tmp_path):
    empty = tmp_path / "empty"
    empty.mkdir()
    rc, *_ = run_table(tmp_path, 5, empty)
    assert rc == 3  # monkey.obj missing
    rc, lights, spheres, _ = run_table(tmp_path, 8, empty)
    assert rc == 0 and len(spheres) == 3 and len(lights) == 1  # the sphere scene reads no OBJ file
    rc, *_ = run_table(tmp_path, 8, empty, skybox=1)
    assert rc == 3  # sky box faces missing, as in the reference (src/scene.cpp:108-116)


def test_scene_table_reads_the_sky_box(tmp_path):
    """Six JPEG faces right/left/top/bottom/front/back.jpg from the data directory."""
    data = tmp_path / "data"
    data.mkdir()
    g = np.load(os.path.join(o.ROOT, "tests", "golden", "images.npz"))
    src = ["ycc_420.jpg", "ycc_422.jpg", "ycc_444.jpg", "grey.jpg", "own_440.jpg", "ycc_420_progressive.jpg"]
    faces = ["right.jpg", "left.jpg", "top.jpg", "bottom.jpg", "front.jpg", "back.jpg"]
    for s, f in zip(src, faces):
        shutil.copy(os.path.join(IMAGES, s), data / f)
    rc, _, _, sizes = run_table(tmp_path, 8, data, skybox=1)
    assert rc == 0
    assert sizes == [(g[s].shape[1], g[s].shape[0]) for s in src]


# ---- on the reference's own files (only where its checkout exists; nothing of it is stored in this repository)
needs_reference = pytest.mark.skipif(not os.path.isdir(REFERENCE_DATA), reason="reference checkout not present")


@needs_reference
def test_reference_texture_decodes_to_the_golden_texels(tmp_path):
    g = np.load(os.path.join(o.ROOT, "tests", "golden", "cube_textured.npz"))
    rc, px = decode(tmp_path, os.path.join(REFERENCE_DATA, "default.png"))
    assert rc == 0 and np.array_equal(px, g["scene.texture0"])


@needs_reference
@pytest.mark.skipif(not os.path.exists(REF_DUMP), reason="compiled reference not present")
@pytest.mark.parametrize("name", ["texture-2x2.png", "right.jpg", "left.jpg", "top.jpg", "bottom.jpg", "front.jpg", "back.jpg"])
def test_reference_images_decode_like_the_compiled_reference(tmp_path, name):
    npy = str(tmp_path / "ref.npy")
    assert subprocess.run([REF_DUMP, "--image", os.path.join(REFERENCE_DATA, name), npy], timeout=120).returncode == 0
    rc, px = decode(tmp_path, os.path.join(REFERENCE_DATA, name))
    assert rc == 0 and np.array_equal(px, np.load(npy))


@needs_reference
@pytest.mark.parametrize("name", sorted(set(SCENE_TYPES) - {"spheres"}))
def test_scene_table_on_reference_data(tmp_path, name):
    """The whole prebuilt scene — meshes, materials, lights — from the reference's data directory against the golden
    scene arrays."""
    g = np.load(os.path.join(o.ROOT, "tests", "golden", name + ".npz"))
    rc, lights, spheres, sizes = run_table(tmp_path, SCENE_TYPES[name], REFERENCE_DATA)
    assert rc == 0
    assert same_bits(lights, g["scene.lights"])
    v, t, table, m = read_meshes(str(tmp_path / "table.rtscene"))
    assert np.array_equal(table[:, :4], g["scene.mesh_table"][:, :4])
    assert np.array_equal(table[:, 4] > 0, g["scene.mesh_table"][:, 4] >= 0)
    assert np.array_equal(t, g["scene.triangles"])
    assert same_bits(v, g["scene.vertices"])
    if name != "single_triangle":  # that entry overrides kd after loading (src/scene.cpp:13)
        assert same_bits(m, g["scene.materials"])
    if "scene.texture0" in g:
        assert sizes and sizes[0] == (g["scene.texture0"].shape[1], g["scene.texture0"].shape[0])
