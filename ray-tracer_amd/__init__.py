"""Host-side plumbing for the MI355X render-loop library.

The product is the C-ABI shared library built from ``csrc/`` (``librt_amd.so``, declared in
``include/rt_amd.h``) and the C++ facade in ``host/``.  This Python package only binds that C ABI with
ctypes so that tests, ``bench.py`` and multi-GPU launches (``torch.distributed``) can drive it; it holds
no rendering logic and has no CPU fallback: loading fails loudly when the HIP library is missing.
"""
from .capi import (  # noqa: F401
    BvhOptions,
    Camera,
    Features,
    Hit,
    Library,
    Light,
    Params,
    Ray,
    RtError,
    SceneData,
    Stats,
    HIT_DTYPE,
    RAY_DTYPE,
    default_params,
    features,
    library_path,
    make_camera,
)
