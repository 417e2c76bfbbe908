#!/usr/bin/env python3
"""Parity sweep (run by hand on a GPU box, not collected by pytest): random frame sizes, sample counts, depths, seeds and
row selections over every built-in scene, the HIP path through the C-ABI against the CPU oracle, bytes and bits.

    python tests/stress_parity.py [seconds=300] [seed=0]        RT_STRESS_BIG=1: larger frames at 16-64 spp (the long
                                                                  pixels of glass scenes reach the hand-over threshold)
                                                                RT_STRESS_LARGE=1: only scenes of 257-900 hittables
                                                                  (the cooperative route), more pixels than waves fit
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib                      # noqa: E402
import raytracer_amd as ra             # noqa: E402
from raytracer_amd import scenes, _abi  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
builders = [scenes.default4, scenes.random10, scenes.glass_mirror, scenes.plane_scene, scenes.cube_scene,
            scenes.point_light_scene, scenes.degenerate_scene, scenes.nan_scene,
            lambda: scenes.many_objects(n=300, n_lights=4, seed=int(rs.randint(1, 1000)))]


def large_scene():
    """many_objects with what the cluster culling of the large-scene kernels has to get right at random: the scene
    scaled about the camera (distances from 1e-7 to 1e14), directional lights with zero components, planes."""
    sc = scenes.many_objects(n=int(rs.randint(257, 900)), n_lights=int(rs.randint(1, 9)), seed=int(rs.randint(1, 1000)))
    if rs.randint(3) == 0:
        sc.lights = []
        for _ in range(int(rs.randint(1, 4))):
            d = rs.choice([0.0, 0.0, 1.0, -1.0, 0.3], size=3)
            d[1] = -1.0 - rs.randint(2)
            sc.dirlight(tuple(float(v) for v in d))
    if rs.randint(4) == 0:
        sc.plane((0.0, float(rs.uniform(-1.0, 0.2)), 0.0), (0.0, 1.0, 0.0), int(rs.randint(4)))
    if rs.randint(2) == 0:
        k = float(rs.choice([1e-7, 1e-4, 1e-2, 30.0, 1e4, 1e8, 1e11, 1e14]))
        p0 = oracle_lib.camera(sc, 64, 36)[0:3].astype(np.float64)
        at = lambda v: tuple(p0 + k * (np.array(v, dtype=np.float64) - p0))
        out = scenes.Scene(name="%s_x%g" % (sc.name, k))
        out.materials = list(sc.materials)
        for t, a, b, r, m in sc.hittables:
            if t == scenes.SPHERE:
                out.sphere(at(a), r * k, m)
            elif t == scenes.CUBE:
                out.cube(at(a), tuple(k * np.array(b, dtype=np.float64)), m)
            else:
                out.plane(at(a), b, m)
        for t, v in sc.lights:
            if t == scenes.POINTLIGHT:
                out.pointlight(at(v))
            else:
                out.dirlight(v)
        sc = out
    return sc


if os.environ.get("RT_STRESS_LARGE"):
    builders = [large_scene]
rt = ra.Renderer(0)
t_end = time.time() + budget
n_cases = n_pixels = 0
while time.time() < t_end:
    sc = builders[rs.randint(len(builders))]()
    W, H = int(rs.randint(1, 400)), int(rs.randint(1, 250))
    samples = int(rs.choice([1, 2, 3, 5, 8, 9, 16, 33, 64, 70]))
    depth = int(rs.choice([0, 1, 2, 5, 10, 12, 20, 64]))
    if len(sc.hittables) > 100:            # keep the oracle's share of the budget small
        W, H, samples, depth = min(W, 96), min(H, 64), min(samples, 8), min(depth, 10)
        if os.environ.get("RT_STRESS_LARGE"):
            W, H, samples = int(rs.randint(64, 200)), int(rs.randint(40, 120)), int(rs.choice([1, 2, 3]))
    pixel_samples = 3.0e6
    if os.environ.get("RT_STRESS_BIG") and len(sc.hittables) <= 100:
        W, H, samples = int(rs.randint(300, 1300)), int(rs.randint(200, 700)), int(rs.choice([16, 32, 64]))
        depth = min(depth, 12)
        pixel_samples = 4.0e7
    if samples * W * H * (2 if depth > 12 else 1) > pixel_samples:
        samples = max(1, int(pixel_samples / (W * H)))
    seed = int(rs.choice([0, 1, 1000, 123456789, 2**31 - 1, 2**32 - 5, 2**40 + 17]))
    row0 = int(rs.randint(0, H))
    rows = int(rs.randint(1, H - row0 + 1))
    flags = int(rs.choice([0, 0, 0, _abi.RT_FLAG_SINGLE_KERNEL, _abi.RT_FLAG_FORCE_SAMPLE_PARALLEL]))
    if rs.randint(3) == 0:
        flags |= _abi.RT_FLAG_SEQUENCE          # the route of frames in flight (one listed launch, hand-over only in frames bound by their direct kernel)
        if rs.randint(2) == 0:
            flags |= _abi.RT_FLAG_FORCE_WAVE_GROUPS     # ... with the listed kernel one wave per workgroup
    want, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, row0=row0, rows=rows, threads=15)
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    got = np.zeros((rows, W, 3), np.uint8)
    got_em = np.zeros((rows, W, 4), np.float32)
    emission = bool(rs.randint(2))
    rt.render_rows(cam, rt.params(W, H, samples, depth, seed, row0=row0, rows=rows, flags=flags), got, got_em if emission else None)
    ok = np.array_equal(got, want) and (not emission or np.array_equal(got_em.view(np.uint32), want_em.view(np.uint32)))
    n_cases += 1
    n_pixels += rows * W
    if not ok:
        print("MISMATCH:", sc.name, W, H, samples, depth, seed, row0, rows, flags, emission,
              "differing pixels", int((got != want).any(axis=-1).sum()), flush=True)
        sys.exit(1)
    if n_cases % 25 == 0:
        print("%d cases, %d pixels, all identical (%.0f s left)" % (n_cases, n_pixels, t_end - time.time()), flush=True)
print("stress parity: %d cases, %d pixels, every image byte-identical and every emission buffer bit-identical" % (n_cases, n_pixels))
