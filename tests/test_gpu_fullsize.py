"""GPU parity at BASELINE.json's full configuration sizes (the fixture-sized cases are in test_gpu_parity.py):
C1 and C2 whole frames against the committed hashes of the real reference's frames (tests/golden/fullsize_hashes.json,
the values SURVEY.md Appendix C names) and against the CPU oracle; C3 (4K / 64 spp), C4 (2 049 hittables, 8 point
lights, 1080p / 32 spp) and C5 (glass / mirror, 1080p / 64 spp / depth 12) on sampled rows against the CPU oracle.
Bar: bytes identical, tallies equal."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib
import raytracer_amd as ra
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THREADS = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))) - 1)


@pytest.fixture(scope="module")
def rt():
    r = ra.Renderer(0)
    yield r
    r.close()


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_whole_frame_matches_the_reference_hashes(rt, name):
    with open(os.path.join(GOLDEN, "fullsize_hashes.json")) as fh:
        g = json.load(fh)[name]
    z = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    W, H, samples, depth, seed = g["width"], g["height"], g["samples"], g["max_depth"], g["seed_base"]
    sc = scenes.default4()
    rt.upload_scene(sc)
    rgb, em, _ = rt.render(W, H, samples, depth, seed, emission=True)
    assert np.array_equal(rgb[::g["row_step"]], z[name + "_rows"])
    assert np.array_equal(em[::g["row_step"]].view(np.uint32), z[name + "_emission_rows"].view(np.uint32))
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == g["sha256_body"]
    ppm = b"P6\n%d %d\n255\n" % (W, H) + rgb.tobytes()
    assert hashlib.sha256(ppm).hexdigest()[:16] == g["sha256_16_ppm"]
    # the frame again, tallying: the reference's query counts
    _, _, cnt = rt.render(W, H, samples, depth, seed, count=True)
    assert {k: cnt[k] for k in g["counters"]} == g["counters"]
    # ... and the CPU oracle agrees on every byte (what the sampled-row tests below lean on)
    want, _, want_cnt = oracle_lib.render(sc, W, H, samples, depth, seed, threads=THREADS, want_emission=False)
    assert np.array_equal(rgb, want)
    assert {k: want_cnt[k] for k in g["counters"]} == g["counters"]


def sampled_rows(rt, sc, W, H, samples, depth, seed, step, row0=0, flags=0):
    n = (H - row0 + step - 1) // step
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    got = np.zeros((n, W, 3), np.uint8)
    p = rt.params(W, H, samples, depth, seed, row0=row0, rows=n, band_rows=1, band_stride=step, flags=flags)
    rt.render_rows(cam, p, got)
    want, _, _ = oracle_lib.render_strided(sc, W, H, samples, depth, seed, row0=row0, rows=n, row_step=step, threads=THREADS)
    return got, want


def test_c3_4k_64spp_every_16th_row(rt):
    got, want = sampled_rows(rt, scenes.default4(), 3840, 2160, 64, 10, 0, 16)
    assert np.array_equal(got, want)


def test_c3_4k_64spp_whole_frame_equals_its_sampled_rows(rt):
    """The whole 4K frame through the production kernels (what bench.py times), its every 16th row against the oracle."""
    sc = scenes.default4()
    rt.upload_scene(sc)
    rgb, _, _ = rt.render(3840, 2160, 64, 10, 0)
    want, _, _ = oracle_lib.render_strided(sc, 3840, 2160, 64, 10, 0, row0=5, rows=135, row_step=16, threads=THREADS)
    assert np.array_equal(rgb[5::16], want)


def test_c4_2049_hittables_8_lights_sampled_rows(rt):
    sc = scenes.many_objects()
    assert len(sc.hittables) == 2049 and len(sc.lights) == 8
    got, want = sampled_rows(rt, sc, 1920, 1080, 32, 10, 0, 400, row0=330)       # rows 330 and 730
    assert np.array_equal(got, want)


def test_c4_scene_whole_small_frame_with_tallies(rt):
    """Every pixel of a small frame of the real C4 scene -- all 33 clusters of the culled large-scene kernels, all 8 point
    lights -- through the production route (direct_kernel<BIG> + coop_kernel), bytes and emission bits against the
    oracle; then the tallying build's query counts against the oracle's (hittablesList.h:17-50 semantics)."""
    sc = scenes.many_objects()
    W, H, samples, depth, seed = 240, 135, 2, 10, 0
    rt.upload_scene(sc)
    rgb, em, _ = rt.render(W, H, samples, depth, seed, emission=True)
    want, want_em, want_cnt = oracle_lib.render(sc, W, H, samples, depth, seed, threads=THREADS)
    assert np.array_equal(rgb, want)
    assert np.array_equal(em.view(np.uint32), want_em.view(np.uint32))
    _, _, cnt = rt.render(W, H, samples, depth, seed, count=True)
    for k in ("primary", "closest", "shadow", "tests", "rand_draws"):
        assert cnt[k] == want_cnt[k], k


def test_c5_glass_1080p_64spp_depth12_sampled_rows(rt):
    got, want = sampled_rows(rt, scenes.glass_mirror(), 1920, 1080, 64, 12, 0, 24, row0=7)
    assert np.array_equal(got, want)
