"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/rt_amd.h declares, and
its host-only entry points (camera, argument validation) behave; no compute call is made without a GPU."""
import ctypes
import json
import os
import re

import numpy as np

import raytracer_amd as ra
from raytracer_amd import _abi
from raytracer_amd.renderer import camera_array

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "rt_amd.h")) as fh:
        header = fh.read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    lib = ctypes.CDLL(_abi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_abi.Camera35f) == 35 * 4
    assert ctypes.sizeof(_abi.Params) == 48 and _abi.Params.seed_base.offset == 16
    assert ctypes.sizeof(_abi.Counters) == 48
    assert ctypes.sizeof(_abi.SceneSoA) == 88


def test_camera_init_matches_reference_bits(golden_dir):
    with open(os.path.join(golden_dir, "camera_bits.json")) as fh:
        cams = json.load(fh)
    for key, bits in cams.items():
        W, H = (int(v) for v in key.split("x"))
        assert [int(b) for b in camera_array(ra.camera35f(60.0, W, H)).view(np.uint32)] == bits, key


def test_host_only_validation():
    lib = _abi.load()
    assert lib.rt_camera_init(None, 60.0, 10.0, 10.0, 0.01, 1000.0) == _abi.RT_ERR_INVALID_ARG
    cam = _abi.Camera35f()
    assert lib.rt_camera_init(ctypes.byref(cam), 60.0, 0.0, 10.0, 0.01, 1000.0) == _abi.RT_ERR_INVALID_ARG
    assert lib.rt_create(None, 0) == _abi.RT_ERR_INVALID_ARG
    assert lib.rt_strerror(_abi.RT_ERR_SCENE).decode().startswith("malformed scene")
    assert lib.rt_render_rows(None, None, None, None, None, None) == _abi.RT_ERR_INVALID_ARG
    assert lib.rt_upload_scene(None, None) == _abi.RT_ERR_INVALID_ARG
    # the multi-GPU entry points reject bad arguments before they touch RCCL or a GPU
    assert lib.rt_gather_rows(None, None, 0, 1, None, None, None, 1, 1, 1, 1, 1) == _abi.RT_ERR_INVALID_ARG
    assert lib.rt_comm_create(None, None, None, 0, 1) == _abi.RT_ERR_INVALID_ARG
    comm = ctypes.c_void_p(123)
    assert lib.rt_comm_create(ctypes.byref(comm), None, None, 0, 1) == _abi.RT_ERR_INVALID_ARG and comm.value is None
    assert lib.rt_comm_unique_id(None) == _abi.RT_ERR_INVALID_ARG
    lib.rt_comm_destroy(None)
    assert lib.rt_last_rccl_error(None) == 0
    assert b"RCCL" in lib.rt_strerror(_abi.RT_ERR_RCCL)


def test_no_silent_cpu_fallback():
    """Without a GPU the product refuses to render (the oracle is never a stand-in)."""
    import torch
    if torch.cuda.is_available():
        return
    try:
        ra.Renderer(0)
    except ra.RtError as e:
        assert e.status == _abi.RT_ERR_NO_DEVICE
    else:
        raise AssertionError("Renderer() succeeded without a GPU")
    for fn in os.listdir(os.path.join(ROOT, "raytracer_amd")):
        if fn.endswith(".py"):
            with open(os.path.join(ROOT, "raytracer_amd", fn)) as fh:
                src = fh.read()
            assert "oracle_lib" not in src and "librt_oracle" not in src, fn
