"""rt_render_rows_static: the entry point that allocates nothing (its per-frame state and pixel classes live in a static
device array).  Called through the C-ABI and compared with rt_render_rows -- byte-identical images, bit-identical
emission -- on two calls in a row, the second with a smaller input (the array keeps the first call's contents), and with
the CPU oracle on a band.  Collected last (file name): `pytest -x` stops at the first failure."""
import ctypes

import numpy as np
import pytest

import oracle_lib
import raytracer_amd as ra
from raytracer_amd import _abi, scenes

pytestmark = pytest.mark.gpu


def _both(rt, sc, W, H, samples, depth, seed, emission=True, **rows):
    import torch
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    p = rt.params(W, H, samples, depth, seed, **rows)
    dev = "cuda:%d" % rt.device
    out = []
    for static in (False, True):
        rgb = torch.full((p.rows, W, 3), 7, dtype=torch.uint8, device=dev)
        em = torch.full((p.rows, W, 4), -1.0, dtype=torch.float32, device=dev) if emission else None
        torch.cuda.synchronize()
        if static:
            rt.render_rows_static(cam, p, rgb, em)
        else:
            rt.render_rows(cam, p, rgb, em)
        out.append((rgb.cpu().numpy(), em.cpu().numpy() if emission else None))
    return out


def test_static_entry_point_equals_render_rows_twice_in_a_row():
    with ra.Renderer(0) as rt:
        # first call: a frame with many bounce pixels (classes + a long pixel list in the scratch array)
        sc = scenes.glass_mirror()
        W, H, samples, depth, seed = 640, 360, 8, 12, 3
        (a, ea), (b, eb) = _both(rt, sc, W, H, samples, depth, seed)
        assert np.array_equal(a, b) and np.array_equal(ea.view(np.uint32), eb.view(np.uint32))
        want, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, row0=150, rows=32, threads=8)
        assert np.array_equal(b[150:182], want) and np.array_equal(eb[150:182].view(np.uint32), want_em.view(np.uint32))

        # second call, smaller input, other scene: nothing of the first call's state may leak into it
        sc = scenes.default4()
        W, H, samples, depth, seed = 200, 77, 5, 10, 11
        (a, ea), (b, eb) = _both(rt, sc, W, H, samples, depth, seed)
        assert np.array_equal(a, b) and np.array_equal(ea.view(np.uint32), eb.view(np.uint32))
        want, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
        assert np.array_equal(b, want) and np.array_equal(eb.view(np.uint32), want_em.view(np.uint32))

        # third: cyclic bands, no emission buffer, a scene with planes and cubes, and a large scene (cooperative route)
        sc = scenes.cube_scene()
        (a, _), (b, _) = _both(rt, sc, 160, 96, 3, 5, 2, emission=False, row0=16, rows=32, band_rows=16, band_stride=48)
        assert np.array_equal(a, b)
        sc = scenes.many_objects(n=300, n_lights=2, seed=23)
        (a, _), (b, _) = _both(rt, sc, 96, 48, 2, 4, 6, emission=False)
        assert np.array_equal(a, b)


def test_static_entry_point_leaves_free_device_memory_unchanged():
    """What the entry point is for: a context whose FIRST render is the static call, then a call with more bounce pixels
    than the one before, then a smaller one -- the device's free memory (hipMemGetInfo) is the same before and after
    each.  (rt_render_rows on the same fresh context does allocate: its pixel classes.)"""
    import torch
    torch.zeros(1, device="cuda:0")
    sc = scenes.glass_mirror()
    with ra.Renderer(0) as rt:
        rt.upload_scene(sc)
        outs = {(W, H): torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0") for (W, H) in ((160, 90), (640, 360), (96, 48))}
        ems = {k: torch.zeros(v.shape[:2] + (4,), dtype=torch.float32, device="cuda:0") for k, v in outs.items()}
        torch.cuda.synchronize()
        deltas = []
        for (W, H), spp in (((160, 90), 4), ((640, 360), 8), ((96, 48), 2)):
            cam = ra.camera35f(sc, W, H)
            p = rt.params(W, H, spp, 12, 1)
            free0, _ = torch.cuda.mem_get_info(0)
            rt.render_rows_static(cam, p, outs[(W, H)], ems[(W, H)])
            free1, _ = torch.cuda.mem_get_info(0)
            deltas.append(free0 - free1)
        assert deltas == [0, 0, 0], deltas
        # the allocating entry point on the same context, for contrast: it has to make room for the pixel classes
        free0, _ = torch.cuda.mem_get_info(0)
        rt.render_rows(ra.camera35f(sc, 640, 360), rt.params(640, 360, 8, 12, 1), outs[(640, 360)], ems[(640, 360)])
        free1, _ = torch.cuda.mem_get_info(0)
        assert free0 - free1 > 0


def test_static_entry_point_limits():
    import torch
    with ra.Renderer(0) as rt:
        sc = scenes.default4()
        rt.upload_scene(sc)
        lib = _abi.load()
        # exactly the maximum passes, one row more is refused with a clear error
        W, rows = 2048, _abi.RT_STATIC_MAX_PIXELS // 2048
        cam = ra.camera35f(sc, W, rows + 1)
        rgb = torch.zeros((rows + 1, W, 3), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        rt.render_rows_static(cam, rt.params(W, rows + 1, 1, 2, 0, rows=rows), rgb)
        ref = torch.zeros_like(rgb)
        torch.cuda.synchronize()
        rt.render_rows(cam, rt.params(W, rows + 1, 1, 2, 0, rows=rows), ref)
        assert torch.equal(rgb, ref)
        with pytest.raises(ra.RtError) as e:
            rt.render_rows_static(cam, rt.params(W, rows + 1, 1, 2, 0), rgb)
        assert e.value.status == _abi.RT_ERR_TOO_LARGE and "RT_STATIC_MAX_PIXELS" in str(e.value)
        # host outputs and tallies are refused
        host = np.zeros((4, 64, 3), np.uint8)
        cam = ra.camera35f(sc, 64, 4)
        with pytest.raises(ra.RtError) as e:
            rt.render_rows_static(cam, rt.params(64, 4, 1, 2, 0), host)
        assert e.value.status == _abi.RT_ERR_INVALID_ARG
        small = torch.zeros((4, 64, 3), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        with pytest.raises(ra.RtError) as e:
            rt.render_rows_static(cam, rt.params(64, 4, 1, 2, 0, flags=_abi.RT_FLAG_COUNT_LIKE_REFERENCE), small)
        assert e.value.status == _abi.RT_ERR_INVALID_ARG
        assert lib.rt_render_rows_static(None, None, None, None, None) == _abi.RT_ERR_INVALID_ARG
