"""ctypes binding of include/rt_amd.h (raytracer_amd/librt_amd.so).  Plumbing only: every pixel comes from the HIP
kernels behind the C-ABI.  There is no CPU fallback; a missing library or GPU raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB lets an A/B experiment point at another build of the same C-ABI
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

RT_OK = 0
RT_ERR_INVALID_ARG, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_SCENE = -1, -2, -3, -4
RT_ERR_UNSUPPORTED, RT_ERR_NO_SCENE, RT_ERR_DEVICE_FAULT, RT_ERR_TOO_LARGE = -5, -6, -7, -8
RT_STATIC_MAX_PIXELS = 1 << 20
RT_FLAG_COUNT_LIKE_REFERENCE = 1
RT_FLAG_COUNT_EXECUTED = 2
RT_FLAG_SINGLE_KERNEL = 4
RT_FLAG_FORCE_SAMPLE_PARALLEL = 8
RT_FLAG_SEQUENCE = 16
RT_FLAG_RELAXED = 128
RT_FLAG_FORCE_WAVE_GROUPS = 256
RT_COMM_ID_BYTES = 128
RT_ERR_RCCL = -9

EXPORTS = [
    "rt_create", "rt_destroy", "rt_upload_scene", "rt_camera_init", "rt_render_rows", "rt_render_rows_static", "rt_render_tiles", "rt_probe_max_depth",
    "rt_deinterleave_bands", "rt_comm_unique_id", "rt_comm_create", "rt_comm_destroy", "rt_gather_rows", "rt_render_gather", "rt_last_rccl_error", "rt_comm_count", "rt_last_gather_ms", "rt_apply_glow",
    "rt_last_kernel_ms", "rt_last_device_ms", "rt_last_hip_error", "rt_strerror",
    "rt_debug_rand", "rt_debug_quantise", "rt_debug_intersect", "rt_debug_cull", "rt_debug_cull_enter", "rt_debug_exact_math",
]


# rt_tile_sink: int (*)(void* user, const uint8_t* record, size_t bytes)
TILE_SINK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_size_t)


class SceneSoA(ctypes.Structure):
    _fields_ = [("n_materials", ctypes.c_int32), ("materials", ctypes.c_void_p),
                ("n_hittables", ctypes.c_int32), ("h_type", ctypes.c_void_p), ("h_a", ctypes.c_void_p),
                ("h_b", ctypes.c_void_p), ("h_radius", ctypes.c_void_p), ("h_material", ctypes.c_void_p),
                ("n_lights", ctypes.c_int32), ("l_type", ctypes.c_void_p), ("l_v", ctypes.c_void_p)]


class Camera35f(ctypes.Structure):
    _fields_ = [("position", ctypes.c_float * 3), ("inv_proj", ctypes.c_float * 16), ("inv_view", ctypes.c_float * 16)]


class Params(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("samples", ctypes.c_int32),
                ("max_depth", ctypes.c_int32), ("seed_base", ctypes.c_int64), ("row0", ctypes.c_int32),
                ("rows", ctypes.c_int32), ("band_rows", ctypes.c_int32), ("band_stride", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


class Counters(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in ("primary", "closest", "shadow", "tests", "rand_draws", "shaded")]


class RtError(RuntimeError):
    def __init__(self, status, what, hip=0):
        self.status, self.hip = status, hip
        super().__init__("%s failed: %s (status %d, %s error %d)" % (what, _strerror(status), status, "rccl" if status == -9 else "hip", hip))


_lib = None


def load():
    """Loads librt_amd.so.  torch (if importable) is imported first so that both share one libamdhip64."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("raytracer_amd: %s is missing -- build it with `make -C raytracer_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads torch's bundled libamdhip64.so.7 before ours resolves the same soname)
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.rt_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    lib.rt_upload_scene.argtypes = [vp, ctypes.POINTER(SceneSoA)]
    lib.rt_camera_init.argtypes = [ctypes.POINTER(Camera35f)] + [ctypes.c_float] * 5
    lib.rt_render_rows.argtypes = [vp, ctypes.POINTER(Camera35f), ctypes.POINTER(Params), vp, vp, vp]
    lib.rt_render_rows_static.argtypes = [vp, ctypes.POINTER(Camera35f), ctypes.POINTER(Params), vp, vp]
    lib.rt_probe_max_depth.argtypes = [vp, ctypes.POINTER(Camera35f), ctypes.POINTER(Params), vp]
    lib.rt_render_tiles.argtypes = [vp, ctypes.POINTER(Camera35f), ctypes.POINTER(Params), i32, i32, TILE_SINK, vp]
    lib.rt_deinterleave_bands.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32]
    lib.rt_comm_unique_id.argtypes = [vp]
    lib.rt_comm_create.argtypes = [ctypes.POINTER(vp), vp, vp, i32, i32]
    lib.rt_comm_destroy.argtypes = [vp]
    lib.rt_comm_destroy.restype = None
    lib.rt_gather_rows.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32]
    lib.rt_last_rccl_error.argtypes = [vp]
    lib.rt_comm_count.argtypes = [vp, ctypes.POINTER(i32)]
    lib.rt_last_gather_ms.argtypes = [vp]
    lib.rt_last_gather_ms.restype = ctypes.c_float
    lib.rt_apply_glow.argtypes = [vp, vp, vp, i32, i32, ctypes.POINTER(i32)]
    lib.rt_render_gather.argtypes = [vp, ctypes.POINTER(Camera35f), ctypes.POINTER(Params), vp, i32, i32, i32, vp]
    lib.rt_last_kernel_ms.argtypes = [vp]
    lib.rt_last_kernel_ms.restype = ctypes.c_float
    lib.rt_last_device_ms.argtypes = [vp]
    lib.rt_last_device_ms.restype = ctypes.c_float
    lib.rt_last_hip_error.argtypes = [vp]
    lib.rt_strerror.argtypes = [ctypes.c_int]
    lib.rt_strerror.restype = ctypes.c_char_p
    lib.rt_debug_rand.argtypes = [vp, vp, i32, i32, vp]
    lib.rt_debug_quantise.argtypes = [vp, vp, i32, vp]
    lib.rt_debug_intersect.argtypes = [vp, i32, vp, i32, vp]
    lib.rt_debug_cull.argtypes = [vp, i32, vp, i32, vp]
    lib.rt_debug_cull_enter.argtypes = [vp, i32, vp, i32, vp]
    lib.rt_debug_exact_math.argtypes = [vp, vp]
    _lib = lib
    return lib


def _strerror(status):
    try:
        return load().rt_strerror(status).decode()
    except Exception:
        return "status %d" % status
