"""The command-line renderer (ray-tracer_amd/host/rt_cli.cpp) and its configuration reader (rt_config.cpp on
toml_lite.h) against the reference's readConfigFile / CLI mode (reference src/config.cpp:71-431, src/main.cpp:499-556):
  - what the configuration printer shows for the TOML files under tests/golden/config/ (written for this
    repository) must be, byte for byte, what the compiled reference printed (oracle/gen_config_golden.py);
  - the keys this build adds, the reader's error cases;
  - on the GPU: frames rendered through the program against the golden images."""
import glob
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as o

CLI = os.path.join(o.ROOT, "ray-tracer_amd", "host", "rt_cli")
CONFIGS = os.path.join(o.ROOT, "tests", "golden", "config")


def run_cli(args, cwd=None, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([CLI] + args, capture_output=True, text=True, timeout=timeout, cwd=cwd or o.ROOT, env=e)


@pytest.mark.parametrize("name", ["full", "minimal", "strings"])
def test_configuration_prints_like_the_reference(name):
    res = run_cli([os.path.join("tests", "golden", "config", name + ".toml"), "--print-config"], env={"RT_DATA_DIR": "<DATA_DIR>"})
    assert res.returncode == 0, res.stderr
    assert res.stdout == open(os.path.join(CONFIGS, name + ".out")).read()
    assert res.stderr == open(os.path.join(CONFIGS, name + ".err")).read()


BASE = """
command_line_rendering = true
window_size = [8, 8]
scene = "spheres"
output_dir = "/tmp/rt_cli_test_out"
[features]
enable_shading = true
enable_recursive = false
enable_hard_shadow = false
enable_normal_interp = false
enable_texture_mapping = false
enable_accel_structure = true
"""


def print_config(tmp_path, text, env=None):
    cfg = tmp_path / "c.toml"
    cfg.write_text(text)
    return run_cli([str(cfg), "--print-config"], env=env)


def test_added_keys_and_fallbacks(tmp_path):
    res = print_config(tmp_path, BASE + "enable_soft_shadow = true\n[features.extra]\nenable_transparency = true\nenable_bvh_sah_binning = true\n")
    assert res.returncode == 0, res.stderr
    assert "enable_soft_shadow: true" in res.stdout and "enable_transparency: true" in res.stdout and "enable_bvh_sah_binning: true" in res.stdout
    # a switch the reference would crash on when absent: its value_or() default, and a warning
    res = print_config(tmp_path, BASE.replace("enable_recursive = false\n", "").replace("command_line_rendering = true\n", ""))
    assert res.returncode == 0
    assert "command_line_rendering: true" in res.stdout and "enable_recursive: false" in res.stdout
    assert "features.enable_recursive is missing" in res.stderr and "command_line_rendering is missing" in res.stderr
    # integers where the reference insists on floats
    res = print_config(tmp_path, BASE + "[[cameras]]\nfield_of_view = 40\ndistance_from_look_at = 2\nlook_at = [1, 2, 3]\n")
    assert "field_of_view: 40" in res.stdout and "look_at: [1, 2, 3]" in res.stdout and "rotation: [0, 0, 0]" in res.stdout
    # data_path from the environment
    res = print_config(tmp_path, BASE, env={"RT_DATA_DIR": "/somewhere/else"})
    assert 'data_path: "/somewhere/else"' in res.stdout


@pytest.mark.parametrize("text,message", [
    ("a = 1\na = 2\n", "defined twice"),
    ("a = [1, 2\n", "expected"),
    ("a = \"unterminated\n", "unterminated string"),
    ("[t]\nx = 1\n[t]\ny = 2\n", "defined twice"),
    ("a = 1979-05-27\n", "dates and times"),
    ("a = 1 2\n", "unexpected text"),
    ("a = 01\n", "leading zero"),
    ("a = { b = 1 }\na.c = 2\n", "inline table"),
    ("a = \"\\q\"\n", "unknown escape"),
    ("= 1\n", "expected a key"),
    pytest.param("a = " + "[" * 5000 + "]" * 5000 + "\n", "nested deeper", id="deep-nesting"),  # refused, no stack overflow
])
def test_malformed_toml_is_refused(tmp_path, text, message):
    res = print_config(tmp_path, text)
    assert res.returncode == 1
    assert "Failed parsing" in res.stderr and message in res.stderr


def test_unknown_scene_and_usage(tmp_path):
    res = print_config(tmp_path, BASE.replace('"spheres"', '"no_such_scene.obj"'))
    assert res.returncode == 1 and "does not exist" in res.stderr
    assert run_cli([]).returncode == 2
    assert run_cli([str(tmp_path / "missing.toml")]).returncode == 1
    cfg = tmp_path / "window.toml"
    cfg.write_text(BASE.replace("command_line_rendering = true", "command_line_rendering = false"))
    res = run_cli([str(cfg)])
    assert res.returncode == 2 and "interactive window" in res.stderr


def read_bmp(path, w, h):
    bmp = np.fromfile(path, dtype=np.uint8)
    assert bmp[0] == ord("B") and bmp[1] == ord("M") and len(bmp) == 54 + w * h * 4
    return bmp[54:].reshape(h, w, 4)[::-1, :, 2::-1]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--tiles", "3"], ["--gather-on-device"]], ids=["whole-frames", "row-tiles", "device-gather"])
def test_prebuilt_scene_through_the_program_matches_the_golden_image(tmp_path, extra):
    """Scene 8 (spheres: no data files), the golden frame size and camera, shading + hard shadows; once a camera per
    call, once every frame dealt in 16-row blocks to three shares (the multi-GPU row tiling, here on one GPU), once
    through the GPU-to-GPU gather (rt_render_blocks_gather on a one-rank RCCL communicator: one GPU is all there is)."""
    W, H = o.SIZE_A
    out = tmp_path / "out"
    cfg = tmp_path / "spheres.toml"
    cfg.write_text(f"""
command_line_rendering = true
window_size = [{W}, {H}]
scene = 8
output_dir = "{out}"
[features]
enable_shading = true
enable_recursive = false
enable_hard_shadow = true
enable_normal_interp = false
enable_texture_mapping = false
enable_accel_structure = true
[[cameras]]
field_of_view = {float(o.CAM_A['fov_deg'])}
distance_from_look_at = {float(o.CAM_A['distance'])}
look_at = [{', '.join(str(float(v)) for v in o.CAM_A['look_at'])}]
rotation = [{', '.join(str(float(v)) for v in o.CAM_A['rot_deg'])}]
[[cameras]]
field_of_view = 70.0
distance_from_look_at = 5.0
look_at = [0.0, 0.0, 6.0]
rotation = [0.0, 180.0, 0.0]
""")
    res = run_cli([str(cfg)] + extra)
    assert res.returncode == 0, res.stderr
    assert "renderRayTracing" not in res.stderr
    assert "2 images rendered." in res.stdout and "SceneType::Spheres" in res.stdout
    files = sorted(glob.glob(str(out / "spheres_*_cam_*.bmp")))
    assert len(files) == 2 and files[0].endswith("_cam_0.bmp")
    px = read_bmp(files[0], W, H).astype(int)
    want = (np.clip(o.golden("spheres")["img.A.hard"], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8).astype(int)
    assert np.abs(px - want).max() <= 1 and (px == want).mean() >= 0.999
    other = read_bmp(files[1], W, H)
    assert other.max() > 0 and not np.array_equal(other, px)


@pytest.mark.gpu
def test_obj_scene_with_configured_lights_and_tunables(tmp_path):
    """A scene file under data_path with the lights of the configuration (loadSceneFromFile), soft shadows and
    supersampling set through the added keys; two runs give the same bytes."""
    out = tmp_path / "out"
    cfg = tmp_path / "objects.toml"
    cfg.write_text(f"""
command_line_rendering = true
window_size = [96, 64]
data_path = "{os.path.join(o.ROOT, 'tests', 'golden', 'obj_own')}"
scene = "objects.obj"
output_dir = "{out}"
[features]
enable_shading = true
enable_recursive = false
enable_hard_shadow = false
enable_soft_shadow = true
enable_normal_interp = true
enable_texture_mapping = true
enable_accel_structure = true
[features.extra]
enable_multiple_rays_per_pixel = true
[tunables]
ray_multiplier = 2
samples_per_unit_parallel = 5
seed = 3
[[cameras]]
field_of_view = 50.0
distance_from_look_at = 4.0
look_at = [0.0, 0.0, 0.0]
rotation = [20.0, 30.0, 0.0]
[[lights]]
type = "parallelogram"
corner = [-0.5, 3.0, -0.5]
edges = [[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]
colors = [[1.0, 1.0, 1.0], [1.0, 0.9, 0.8], [0.8, 0.9, 1.0], [1.0, 1.0, 1.0]]
""")
    frames = []
    for _ in range(2):
        res = run_cli([str(cfg), "--gpus", "1"])
        assert res.returncode == 0, res.stderr
        assert "renderRayTracing:" not in res.stderr
        files = sorted(glob.glob(str(out / "objects_*_cam_0.bmp")), key=os.path.getmtime)
        frames.append(read_bmp(files[-1], 96, 64))
    assert np.array_equal(frames[0], frames[1])
    lit = frames[0].reshape(-1, 3).max(axis=1)
    assert (lit > 0).mean() > 0.05  # the objects are in view and lit
    # the same picture over the hierarchy of the SAH + binning toggle (built on the GPU by the facade's BvhInterface):
    # another tree, the same closest hits
    cfg.write_text(cfg.read_text().replace("[features.extra]\n", "[features.extra]\nenable_bvh_sah_binning = true\n"))
    res = run_cli([str(cfg), "--gpus", "1"])
    assert res.returncode == 0, res.stderr
    files = sorted(glob.glob(str(out / "objects_*_cam_0.bmp")), key=os.path.getmtime)
    sah = read_bmp(files[-1], 96, 64)
    assert np.abs(sah.astype(int) - frames[0].astype(int)).max() <= 1 and (sah == frames[0]).mean() >= 0.999
