"""rt_gather_blocks / rt_comm_*: frame assembly across GPUs under the C ABI (RCCL loaded at run time).  One GPU is all
a test box has, so what runs here is the single-rank communicator: bootstrap, the grouped call sequence, and the root's
own blocks landing where they belong; the send / receive pairs of more ranks are the same calls with other peers."""
import numpy as np
import pytest
import torch

import oracle_lib as o

rt = o.rt
pytestmark = pytest.mark.gpu


def test_single_rank_gather_assembles_the_frame():
    tiling = __import__("importlib").import_module("ray-tracer_amd.tiling")
    lib = rt.Library()
    lib.init(0)
    g = o.golden("monkey")
    sc = rt.SceneData.from_golden(g)
    bvh = lib.bvh_build(lib.scene_create(sc))
    W, H = 72, 50  # 4 blocks of 16 rows, the last one short
    cam = rt.make_camera(width=W, height=H, **o.CAM_A)
    lights, n = sc.light_array()
    feats, params = rt.features(shading=1, hard_shadow=1), rt.default_params()
    whole, _ = lib.render(bvh, cam, lights, n, feats, params, W, H)
    local = tiling.local_buffer(1, W, H, torch.device("cuda", 0))
    stream = torch.cuda.current_stream().cuda_stream
    lib.render_blocks(bvh, cam, lights, n, feats, params, W, H, tiling.ROW_BLOCK, 0, 1, out=local.data_ptr(), device_ptr=1, stream=stream)
    comm = tiling.native_comm(lib, 0, 1)
    try:
        frame = tiling.gather_frame_native(lib, comm, local, 0, 1, W, H, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), whole.reshape(H, W, 3))
        # bad arguments are refused, not crashed on
        with pytest.raises(rt.RtError):
            lib.gather_blocks(comm, 1, local.data_ptr(), frame.data_ptr(), W, H, tiling.ROW_BLOCK, stream)
        with pytest.raises(rt.RtError):
            lib.gather_blocks(comm, 0, local.data_ptr(), None, W, H, tiling.ROW_BLOCK, stream)
    finally:
        lib.comm_destroy(comm)
