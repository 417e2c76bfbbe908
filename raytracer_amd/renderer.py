"""Renderer: thin object wrapper over the C-ABI (rt_create / rt_upload_scene / rt_render_rows / ...)."""
import ctypes

import numpy as np

from . import _abi
from .scenes import Scene


def camera35f(scene_or_fov=60.0, width=1024, height=512, znear=0.01, zfar=1000.0):
    """rt_camera_init: Camera(fov, W, H, near, far) of the reference (src/camera.h:11-22) -> 35 float32."""
    if isinstance(scene_or_fov, Scene):
        fov, znear, zfar = scene_or_fov.fov, scene_or_fov.znear, scene_or_fov.zfar
    else:
        fov = scene_or_fov
    cam = _abi.Camera35f()
    rc = _abi.load().rt_camera_init(ctypes.byref(cam), fov, float(width), float(height), znear, zfar)
    if rc != _abi.RT_OK:
        raise _abi.RtError(rc, "rt_camera_init")
    return cam


def camera_array(cam):
    return np.frombuffer(bytes(cam), dtype=np.float32).copy()


def _ptr(buf):
    """host numpy array or torch tensor (host or device) -> raw address"""
    if buf is None:
        return None
    if isinstance(buf, np.ndarray):
        assert buf.flags["C_CONTIGUOUS"]
        return buf.ctypes.data
    if hasattr(buf, "data_ptr"):
        assert buf.is_contiguous()
        if buf.is_cuda:
            # the library works on its own HIP stream and its calls are synchronous: work torch still has in flight
            # on this tensor (its zero fill, a collective) must have landed before the library touches it
            import torch
            torch.cuda.current_stream(buf.device).synchronize()
        return buf.data_ptr()
    raise TypeError("expected a numpy array or a torch tensor")


class Renderer:
    """One context on one GPU (not thread-safe, synchronous calls), mirroring the C-ABI one to one."""

    def __init__(self, device=0):
        self._lib = _abi.load()
        h = ctypes.c_void_p()
        rc = self._lib.rt_create(ctypes.byref(h), int(device))
        if rc != _abi.RT_OK:
            raise _abi.RtError(rc, "rt_create")
        self._h = h
        self.device = int(device)
        self.scene = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != _abi.RT_OK:
            raise _abi.RtError(rc, what, self._lib.rt_last_rccl_error(self._h) if rc == _abi.RT_ERR_RCCL else self._lib.rt_last_hip_error(self._h))

    def upload_scene(self, scene: Scene):
        mats, htype, ha, hb, hr, hm, ltype, lv = (np.ascontiguousarray(a) for a in scene.arrays())
        soa = _abi.SceneSoA(len(mats), mats.ctypes.data, len(htype), htype.ctypes.data, ha.ctypes.data, hb.ctypes.data,
                            hr.ctypes.data, hm.ctypes.data, len(ltype), ltype.ctypes.data, lv.ctypes.data)
        self._check(self._lib.rt_upload_scene(self._h, ctypes.byref(soa)), "rt_upload_scene")
        self.scene = scene

    def params(self, width, height, samples, max_depth, seed_base=0, row0=0, rows=None, band_rows=0, band_stride=0, flags=0):
        rows = height - row0 if rows is None else rows
        return _abi.Params(width, height, samples, max_depth, seed_base, row0, rows, band_rows, band_stride, flags)

    def render_rows(self, cam, params, rgb_out=None, emission_out=None, count=False, executed=False):
        """rt_render_rows into caller buffers (numpy = host, torch cuda tensor = device).  -> counters dict or None"""
        cnt = _abi.Counters()
        if count:
            params = _abi.Params.from_buffer_copy(params)
            params.flags |= _abi.RT_FLAG_COUNT_EXECUTED if executed else _abi.RT_FLAG_COUNT_LIKE_REFERENCE
        rc = self._lib.rt_render_rows(self._h, ctypes.byref(cam), ctypes.byref(params), _ptr(rgb_out), _ptr(emission_out),
                                      ctypes.byref(cnt) if count else None)
        self._check(rc, "rt_render_rows")
        if count:
            return {k: int(getattr(cnt, k)) for k in ("primary", "closest", "shadow", "tests", "rand_draws", "shaded")}
        return None

    def render_rows_static(self, cam, params, rgb_dev, emission_dev=None):
        """rt_render_rows_static: the variant that allocates nothing (device outputs only, at most RT_STATIC_MAX_PIXELS
        pixels per call, one call at a time per GPU)."""
        rc = self._lib.rt_render_rows_static(self._h, ctypes.byref(cam), ctypes.byref(params), _ptr(rgb_dev), _ptr(emission_dev))
        self._check(rc, "rt_render_rows_static")

    def render(self, width, height, samples, max_depth, seed_base=0, emission=False, count=False, cam=None, **rows):
        """Convenience: whole frame (or the rows selected by row0/rows/band_rows/band_stride) into fresh host arrays.
        -> (rgb uint8 [rows, W, 3], emission float32 [rows, W, 4] or None, counters or None)"""
        p = self.params(width, height, samples, max_depth, seed_base, **rows)
        cam = cam or camera35f(self.scene, width, height)
        rgb = np.zeros((p.rows, width, 3), dtype=np.uint8)
        em = np.zeros((p.rows, width, 4), dtype=np.float32) if emission else None
        cnt = self.render_rows(cam, p, rgb, em, count=count)
        return rgb, em, cnt

    def render_tiles(self, width, height, samples, max_depth, seed_base=0, tile=16, rows_per_chunk=0, sink=None, cam=None):
        """rt_render_tiles: streams the frame as the reference's Redis tile records (src/redis.h:27-36).  `sink(record:
        bytes)` is called per record (return a true value to stop); without a sink the records are collected.
        -> list of records (bytes) when no sink was given"""
        cam = cam or camera35f(self.scene, width, height)
        p = self.params(width, height, samples, max_depth, seed_base)
        got = []

        def _sink(_user, rec, n):
            data = ctypes.string_at(rec, n)
            if sink is None:
                got.append(data)
                return 0
            return 1 if sink(data) else 0

        cb = _abi.TILE_SINK(_sink)
        self._check(self._lib.rt_render_tiles(self._h, ctypes.byref(cam), ctypes.byref(p), tile, rows_per_chunk, cb, None), "rt_render_tiles")
        return got if sink is None else None

    def probe_max_depth(self, width, height, max_depth, seed_base=0, cam=None):
        p = self.params(width, height, 1, max_depth, seed_base)
        cam = cam or camera35f(self.scene, width, height)
        flags = np.zeros((height, width), dtype=np.uint8)
        self._check(self._lib.rt_probe_max_depth(self._h, ctypes.byref(cam), ctypes.byref(p), flags.ctypes.data), "rt_probe_max_depth")
        return flags

    def deinterleave_bands(self, gathered, frame_out, width, height, bytes_per_pixel, band_rows, nranks, rows_per_rank):
        self._check(self._lib.rt_deinterleave_bands(self._h, _ptr(gathered), _ptr(frame_out), width, height, bytes_per_pixel,
                                                    band_rows, nranks, rows_per_rank), "rt_deinterleave_bands")

    # multi-GPU: the band gather over RCCL (include/rt_amd.h rt_gather_rows); communicators are plain ncclComm_t handles
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId through the library (one rank calls it; the bytes go to the other ranks by any means)."""
        buf = (ctypes.c_ubyte * _abi.RT_COMM_ID_BYTES)()
        rc = _abi.load().rt_comm_unique_id(buf)
        if rc != 0:
            raise _abi.RtError(rc, "rt_comm_unique_id")
        return bytes(buf)

    def comm_create(self, unique_id, rank, nranks):
        """ncclCommInitRank on this context's GPU.  -> communicator handle (int) for gather_rows / comm_destroy"""
        if len(unique_id) != _abi.RT_COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _abi.RT_COMM_ID_BYTES)
        comm = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * _abi.RT_COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.rt_comm_create(ctypes.byref(comm), self._h, buf, rank, nranks), "rt_comm_create")
        return comm.value

    def comm_destroy(self, comm):
        if comm:
            self._lib.rt_comm_destroy(ctypes.c_void_p(comm))

    @staticmethod
    def comm_count(comm):
        """ncclCommCount: the number of ranks the communicator spans, as RCCL sees it"""
        n = ctypes.c_int32(0)
        rc = _abi.load().rt_comm_count(ctypes.c_void_p(comm), ctypes.byref(n))
        if rc != 0:
            raise _abi.RtError(rc, "rt_comm_count")
        return int(n.value)

    @property
    def last_gather_ms(self):
        """device time of this context's last gather_rows (RCCL group + reassembly), from HIP events on its stream"""
        return float(self._lib.rt_last_gather_ms(self._h))

    def gather_rows(self, comm, rank, nranks, block, gathered, frame_out, width, height, bytes_per_pixel, band_rows, rows_per_rank):
        """rt_gather_rows: this rank's packed band block to rank 0 (grouped ncclSend / ncclRecv on the context's stream),
        reassembled into frame_out there.  block / gathered / frame_out: device tensors (gathered, frame_out: rank 0)."""
        self._check(self._lib.rt_gather_rows(self._h, ctypes.c_void_p(comm) if comm else None, rank, nranks, _ptr(block), _ptr(gathered),
                                             _ptr(frame_out), width, height, bytes_per_pixel, band_rows, rows_per_rank), "rt_gather_rows")

    def apply_glow(self, rgb, emission, width, height):
        """rt_apply_glow: the reference's applyGlow() (src/main.cpp:163-229) on rgb [H, W, 3] uint8 in place, from emission
        [H, W, 4] float32 (numpy = host, torch cuda tensors = device).  -> pyramid levels applied"""
        levels = ctypes.c_int32(0)
        self._check(self._lib.rt_apply_glow(self._h, _ptr(rgb), _ptr(emission), width, height, ctypes.byref(levels)), "rt_apply_glow")
        return int(levels.value)

    @property
    def last_kernel_ms(self):
        return float(self._lib.rt_last_kernel_ms(self._h))

    @property
    def last_device_ms(self):
        return float(self._lib.rt_last_device_ms(self._h))

    # unit-test hooks
    def debug_rand(self, seeds, n_draws):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        out = np.zeros((len(seeds), n_draws), dtype=np.int32)
        self._check(self._lib.rt_debug_rand(self._h, seeds.ctypes.data, len(seeds), n_draws, out.ctypes.data), "rt_debug_rand")
        return out

    def debug_quantise(self, values):
        values = np.ascontiguousarray(values, dtype=np.float32)
        out = np.zeros(len(values), dtype=np.uint8)
        self._check(self._lib.rt_debug_quantise(self._h, values.ctypes.data, len(values), out.ctypes.data), "rt_debug_quantise")
        return out

    def debug_intersect(self, index, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 8), dtype=np.float32)
        self._check(self._lib.rt_debug_intersect(self._h, index, rays.ctypes.data, len(rays), out.ctypes.data), "rt_debug_intersect")
        return out

    def debug_exact_math(self):
        """inputs (of all 2^32 floats) on which the kernels' short sqrt / reciprocal / 1/sqrt differ from IEEE: (0, 0, 0)"""
        out = np.zeros(3, dtype=np.uint64)
        self._check(self._lib.rt_debug_exact_math(self._h, out.ctypes.data), "rt_debug_exact_math")
        return tuple(int(v) for v in out)

    def debug_cull(self, index, rays):
        """1 per ray if the large-scene kernels would visit hittable `index`'s cluster, 0 if they would skip it"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.zeros(len(rays), dtype=np.uint8)
        self._check(self._lib.rt_debug_cull(self._h, index, rays.ctypes.data, len(rays), out.ctypes.data), "rt_debug_cull")
        return out

    def debug_cull_enter(self, index, rays):
        """per ray the parameter at which it enters hittable `index`'s cluster box as the distance rule of the large-scene
        kernels sees it (-FLT_MAX where the rule does not apply)"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.zeros(len(rays), dtype=np.float32)
        self._check(self._lib.rt_debug_cull_enter(self._h, index, rays.ctypes.data, len(rays), out.ctypes.data), "rt_debug_cull_enter")
        return out


def write_ppm(path, rgb):
    """P6 writer with the reference's header (src/main.cpp:423)."""
    h, w, _ = rgb.shape
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (w, h))
        fh.write(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())
