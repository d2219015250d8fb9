"""rt_bloom_static, the entry point that allocates nothing (its frame copy lives in a static device array of the
library), against rt_bloom on the same device frames: bit-identical, on two calls in a row, the second with a
smaller frame; frames beyond the static array are refused."""
import numpy as np
import pytest

import oracle_lib as o

rt = o.rt
pytestmark = pytest.mark.gpu


def _frame(rng, h, w):
    # a frame with pixels on both sides of the bloom threshold, including exact threshold values
    img = rng.random((h, w, 3), dtype=np.float32) * np.float32(1.4)
    img[::7, ::5] = np.float32(0.7)
    return img


def test_bloom_static_matches_bloom_on_two_calls_in_a_row():
    import torch

    lib = rt.Library()
    lib.init(0)
    rng = np.random.default_rng(5)
    stream = torch.cuda.current_stream().cuda_stream
    cases = [((270, 480), rt.default_params()),
             ((50, 33), rt.default_params(bloom_gaussian=0, bloom_size=3, bloom_threshold=0.5, bloom_scale=1.5)),
             ((64, 64), rt.default_params(bloom_debug=1, bloom_size=1, bloom_sigma=0.5))]
    for (h, w), params in cases:  # every call after the first reuses the array the call before it filled
        img = _frame(rng, h, w)
        a = torch.from_numpy(img).cuda()
        b = a.clone()
        lib.bloom_device(a.data_ptr(), w, h, params, stream)
        lib.bloom_static(b.data_ptr(), w, h, params, stream)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        ref = o.bloom(img, params)
        assert np.abs(a.cpu().numpy() - ref).max() <= 2e-6
        assert not np.array_equal(a.cpu().numpy(), img)


def test_bloom_static_refuses_frames_beyond_its_array():
    import torch

    lib = rt.Library()
    lib.init(0)
    assert lib.lib.rt_bloom_static_max_pixels() == 3840 * 2160
    x = torch.zeros(16, device="cuda")
    with pytest.raises(rt.RtError, match="exceeds the static scratch"):
        lib.bloom_static(x.data_ptr(), 3840, 2161, rt.default_params())
