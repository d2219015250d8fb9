"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol the header
declares, and its host BVH builder reproduces the reference's node order (no GPU needed: device copies
are made lazily on the first device call)."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

import golden_cases as gc
import oracle_lib as o

rt = o.rt
ROOT = o.ROOT


@pytest.fixture(scope="module")
def lib():
    return rt.Library()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    assert len(names) >= 15
    for n in sorted(names):
        assert hasattr(lib.lib, n), f"{n} is declared in rt_amd.h but not exported"


def test_struct_sizes_match_header(lib):
    # sizes computed by hand from include/rt_amd.h
    capi = __import__("importlib").import_module("ray-tracer_amd.capi")
    assert C.sizeof(capi.Features) == 17 * 4
    assert C.sizeof(capi.Params) == 21 * 4
    assert C.sizeof(capi.Light) == 22 * 4
    assert C.sizeof(capi.Camera) == 9 * 4
    assert C.sizeof(capi.MeshDesc) == 64
    assert C.sizeof(capi.Stats) == 152  # 7 x u64 + 2 x f32, 2 x u64 + 4 x f32, 2 x u64, 4 x u64, f32 + padding
    p = capi.Params()
    lib.lib.rt_default_params(C.byref(p))
    d = rt.default_params()
    for name, _ in capi.Params._fields_:
        assert getattr(p, name) == getattr(d, name), name


def _export_as_fixture_layout(lib, bvh):
    nodes, prims = lib.bvh_export(bvh)
    meta, box, ids = [], [], []
    for nd in nodes:
        if nd["is_leaf"]:
            cnt = int(nd["b"])
            meta.append((1, nd["level"], len(ids), 2 * cnt))
            for k in range(cnt):
                pr = prims[int(nd["a"]) + k]
                ids += [int(pr["triangle"]), int(pr["mesh"])]
        else:
            meta.append((0, nd["level"], len(ids), 2))
            ids += [int(nd["a"]), int(nd["b"])]
        box.append(list(nd["lower"]) + list(nd["upper"]))
    return np.array(meta, np.int32).reshape(-1, 4), np.array(box, np.float32).reshape(-1, 6), np.array(ids, np.int32)


@pytest.mark.parametrize("scene", [s for s in gc.SCENES])
def test_host_bvh_matches_reference_order(lib, scene):
    g = o.golden(scene)
    sc = rt.SceneData.from_golden(g)
    sh = lib.scene_create(sc)
    bvh = lib.bvh_build(sh)
    try:
        assert [lib.lib.rt_bvh_num_levels(bvh), lib.lib.rt_bvh_num_leaves(bvh)] == list(g["bvh.levels_leaves"])
        meta, box, ids = _export_as_fixture_layout(lib, bvh)
        assert np.array_equal(meta, g["bvh.node_meta"])
        assert np.array_equal(box, g["bvh.node_box"])
        assert np.array_equal(ids, g["bvh.node_ids"])
    finally:
        lib.lib.rt_bvh_destroy(bvh)
        lib.lib.rt_scene_destroy(sh)


@pytest.mark.parametrize("scene", ["cornell", "monkey"])
def test_host_bvh_sah_matches_reference_order(lib, scene):
    g = o.golden(scene)
    sc = rt.SceneData.from_golden(g)
    sh = lib.scene_create(sc)
    bvh = lib.bvh_build(sh, sah=True)
    try:
        assert [lib.lib.rt_bvh_num_levels(bvh), lib.lib.rt_bvh_num_leaves(bvh)] == list(g["bvh_sah.levels_leaves"])
        meta, box, ids = _export_as_fixture_layout(lib, bvh)
        assert np.array_equal(meta, g["bvh_sah.node_meta"])
        assert np.array_equal(box, g["bvh_sah.node_box"])
        assert np.array_equal(ids, g["bvh_sah.node_ids"])
    finally:
        lib.lib.rt_bvh_destroy(bvh)
        lib.lib.rt_scene_destroy(sh)


@pytest.mark.parametrize("sah", [False, True])
def test_host_bvh_does_not_depend_on_the_number_of_build_threads(lib, sah, monkeypatch):
    """Subtrees are built by several host threads and put together in the reference's node order: one thread (the
    reference's own sequence of steps) and eight must give the same bytes."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    sc = rt.SceneData([synth.dragon_class_mesh(5)], lights=synth.area_light())  # 20,480 triangles
    exports = []
    for threads in ("1", "8"):
        monkeypatch.setenv("RT_BUILD_THREADS", threads)
        sh = lib.scene_create(sc)
        bvh = lib.bvh_build(sh, sah=sah)
        nodes, prims = lib.bvh_export(bvh)
        exports.append((nodes.tobytes(), prims.tobytes(), lib.lib.rt_bvh_num_levels(bvh), lib.lib.rt_bvh_num_leaves(bvh)))
        lib.lib.rt_bvh_destroy(bvh)
        lib.lib.rt_scene_destroy(sh)
    assert exports[0] == exports[1]
    assert exports[0][3] == 20480


def test_leaf_size_cap_and_empty_scene(lib):
    g = o.golden("monkey")
    sc = rt.SceneData.from_golden(g)
    sh = lib.scene_create(sc)
    bvh = lib.bvh_build(sh, max_level=3)  # leaves at level 3 hold many triangles
    nodes, prims = lib.bvh_export(bvh)
    assert lib.lib.rt_bvh_num_levels(bvh) == 4
    assert nodes[nodes["is_leaf"] == 1]["b"].sum() == sc.n_triangles == len(prims)
    osc = o.OracleScene(sc, max_level=3)
    assert osc.levels_leaves() == (4, lib.lib.rt_bvh_num_leaves(bvh))
    lib.lib.rt_bvh_destroy(bvh)
    lib.lib.rt_scene_destroy(sh)
    empty = rt.SceneData([])
    sh = lib.scene_create(empty)
    bvh = lib.bvh_build(sh)
    assert lib.lib.rt_bvh_num_nodes(bvh) == 1 and lib.lib.rt_bvh_num_leaves(bvh) == 1
    lib.lib.rt_bvh_destroy(bvh)
    lib.lib.rt_scene_destroy(sh)


def test_argument_errors(lib):
    h = C.c_void_p()
    assert lib.lib.rt_scene_create(None, C.byref(h)) == 1
    assert b"null" in lib.lib.rt_last_error()
    g = o.golden("single_triangle")
    sc = rt.SceneData.from_golden(g)
    sc.meshes[0]["triangles"][0, 0] = 99  # index beyond the vertex count
    sc._build_desc()
    assert lib.lib.rt_scene_create(C.byref(sc.desc), C.byref(h)) == 1
    assert b"beyond" in lib.lib.rt_last_error()
    sc = rt.SceneData.from_golden(g)
    sh = lib.scene_create(sc)
    opt = __import__("importlib").import_module("ray-tracer_amd.capi").BvhOptions(0, 40, 64)
    assert lib.lib.rt_bvh_build(sh, C.byref(opt), C.byref(h)) == 4
    lib.lib.rt_scene_destroy(sh)


def test_gpu_only_entry_points_fail_cleanly_without_a_gpu(lib):
    """rt_bvh_build_device and the frame-assembly calls need a GPU (and RCCL): without one they return a status and a
    message, they do not crash; bad arguments are refused before anything is touched."""
    import torch

    assert lib.lib.rt_gather_blocks(None, 0, None, None, 4, 4, 16, None) == 1  # RT_ERR_INVALID
    assert b"bad argument" in lib.lib.rt_last_error()
    assert lib.lib.rt_comm_init_rank(None, 0, 1, None) == 1
    assert lib.lib.rt_render_blocks_gather(None, None, None, 0, None, None, 4, 4, 16, None, 0, None, 0, None, None) == 1
    lib.lib.rt_comm_destroy(None)
    sc = rt.SceneData.from_golden(o.golden("cornell"))
    for build in (lib.bvh_build, lib.bvh_build_device):  # the same refusals from both builders
        with pytest.raises(rt.RtError, match="sah_bins must be positive"):
            build(lib.scene_create(sc), sah=True, sah_bins=0)
        with pytest.raises(rt.RtError, match="max_level"):
            build(lib.scene_create(sc), max_level=33)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU tests cover the working paths")
    with pytest.raises(rt.RtError):
        lib.bvh_build_device(lib.scene_create(sc))
