"""GPU parity: the HIP path through the C-ABI against (1) the committed golden fixtures produced by the real
reference and (2) the CPU oracle on the same seeded inputs.  Bar: images byte-identical, emission buffers
bit-identical, query counters equal (float32 arithmetic is reproduced operation for operation, so the north-star
tolerance of >=99.9 % pixels within 1 LSB / >=50 dB PSNR is met with zero differing pixels)."""
import glob
import os

import numpy as np
import pytest

import oracle_lib
import raytracer_amd as ra
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith(("probe_", "fullsize")))


@pytest.fixture(scope="module")
def rt():
    r = ra.Renderer(0)
    yield r
    r.close()


def load_case(path):
    z = np.load(path)
    sc = scenes.Scene.from_text(bytes(z["scene"]).decode())
    W, H, samples, depth, seed = (int(v) for v in z["params"])
    return z, sc, W, H, samples, depth, seed


def report(name, got, want):
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32)).max(axis=-1)
    n_bad = int((diff > 1).sum())
    mse = float(((got.astype(np.float64) - want.astype(np.float64)) ** 2).mean())
    psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    print("%s: differing pixels %d, >1 LSB %d of %d, PSNR %.1f dB" % (name, int((diff > 0).sum()), n_bad, diff.size, psnr))
    return n_bad, psnr


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_matches_reference_golden(rt, path):
    z, sc, W, H, samples, depth, seed = load_case(path)
    rt.upload_scene(sc)
    rgb, em, _ = rt.render(W, H, samples, depth, seed, emission=True)
    report(os.path.basename(path), rgb, z["rgb"])
    assert np.array_equal(rgb, z["rgb"])
    assert np.array_equal(em.view(np.uint32), z["emission"].view(np.uint32))
    # the rgb-only kernel variant (no emission bookkeeping) must give the same image
    rgb2, _, _ = rt.render(W, H, samples, depth, seed, emission=False)
    assert np.array_equal(rgb2, z["rgb"])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_counters_match_reference(rt, path):
    z, sc, W, H, samples, depth, seed = load_case(path)
    rt.upload_scene(sc)
    rgb, em, cnt = rt.render(W, H, samples, depth, seed, emission=True, count=True)
    assert [cnt[k] for k in ("primary", "closest", "shadow", "tests", "rand_draws")] == [int(v) for v in z["counters"]]
    assert np.array_equal(rgb, z["rgb"]) and np.array_equal(em.view(np.uint32), z["emission"].view(np.uint32))


def test_camera_bits_match_reference(golden_dir):
    import json
    from raytracer_amd.renderer import camera_array
    with open(os.path.join(golden_dir, "camera_bits.json")) as fh:
        cams = json.load(fh)
    for key, bits in cams.items():
        W, H = (int(v) for v in key.split("x"))
        assert [int(b) for b in camera_array(ra.camera35f(60.0, W, H)).view(np.uint32)] == bits, key


@pytest.mark.parametrize("name,W,H,samples,depth,seed", [
    ("default4", 333, 77, 3, 10, 123456789),        # ragged: width and height not multiples of the 16x16 tile
    ("default4", 17, 1, 2, 10, 0),                  # one row
    ("default4", 1, 19, 2, 4, 5),                   # one column
    ("glass_mirror", 200, 120, 3, 5, 77),
    ("glass_mirror", 64, 64, 2, 20, 1),             # deep stack variant (max_depth > 14)
    ("degenerate", 200, 120, 3, 64, 77),            # RT_MAX_DEPTH
    ("random10", 256, 128, 4, 12, 0),
    ("cube", 200, 120, 3, 5, 77),
    ("plane", 200, 120, 3, 5, 77),
    ("pointlights", 96, 64, 2, 6, 5),
])
def test_matches_oracle_on_seeded_inputs(rt, name, W, H, samples, depth, seed):
    sc = scenes.BUILTIN[name]()
    want_rgb, want_em, want_cnt = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
    rt.upload_scene(sc)
    rgb, em, cnt = rt.render(W, H, samples, depth, seed, emission=True, count=True)
    report(name, rgb, want_rgb)
    assert np.array_equal(rgb, want_rgb)
    assert np.array_equal(em.view(np.uint32), want_em.view(np.uint32))
    assert {k: cnt[k] for k in want_cnt} == want_cnt
    rgb2, _, _ = rt.render(W, H, samples, depth, seed)
    assert np.array_equal(rgb2, want_rgb)


def test_many_objects_small(rt):
    sc = scenes.many_objects(n=300, n_lights=8, seed=11)
    W, H, samples, depth, seed = 96, 48, 1, 5, 2
    want_rgb, want_em, want_cnt = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
    rt.upload_scene(sc)
    rgb, em, cnt = rt.render(W, H, samples, depth, seed, emission=True, count=True)
    assert np.array_equal(rgb, want_rgb) and np.array_equal(em.view(np.uint32), want_em.view(np.uint32)) and {k: cnt[k] for k in want_cnt} == want_cnt


def test_large_scene_cooperative_route(rt):
    """Scenes with >= 256 hittables render their bounce pixels one wave per pixel, the lanes sharing every scan of the
    scene (coop_kernel).  Same bytes as the oracle; a NaN-radius sphere forces the sequential fallback of the
    cooperative closest-hit query (the order of replacements matters once a distance is NaN)."""
    sc = scenes.many_objects(n=300, n_lights=8, seed=11)
    W, H, samples, depth, seed = 128, 72, 3, 8, 4
    want_rgb, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
    rt.upload_scene(sc)
    rgb, em, _ = rt.render(W, H, samples, depth, seed, emission=True)
    assert np.array_equal(rgb, want_rgb) and np.array_equal(em.view(np.uint32), want_em.view(np.uint32))
    rgb, _, _ = rt.render(W, H, samples, depth, seed)
    assert np.array_equal(rgb, want_rgb)

    sc = scenes.many_objects(n=280, n_lights=2, seed=5)
    sc.sphere((1.0, 1.0, -6.0), float("nan"), 2)
    sc.sphere((-2.0, 0.7, -3.0), 0.7, 3)
    want_rgb, want_em, _ = oracle_lib.render(sc, 96, 54, 2, 6, 1, threads=8)
    rt.upload_scene(sc)
    rgb, em, _ = rt.render(96, 54, 2, 6, 1, emission=True)
    assert np.array_equal(rgb, want_rgb) and np.array_equal(em.view(np.uint32), want_em.view(np.uint32))


def _scaled_about_camera(sc, k):
    """The scene scaled by k about the camera's position (the reference's camera does not move): the same view, with
    every distance the primitive tests see multiplied by k."""
    p0 = oracle_lib.camera(sc, 64, 36)[0:3].astype(np.float64)
    out = scenes.Scene(name="%s_x%g" % (sc.name, k))
    out.materials = list(sc.materials)
    out.fov, out.znear, out.zfar = sc.fov, sc.znear, sc.zfar
    at = lambda v: tuple(p0 + k * (np.array(v, dtype=np.float64) - p0))
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
    return out


def test_large_scene_culling_edge_cases(rt):
    """The large-scene kernels skip whole groups of 64 hittables when the ray certainly misses the group's box
    (rt_device.hpp cluster_mask).  Cases at the edges of that argument, each against the oracle, bytes and bits:
    axis-aligned light directions (zero components: 1 / 0 in cube.h:15), planes and NaN / inverted / zero-size
    hittables among the 300 (unbounded or degenerate boxes), and the scene scaled about the camera up to where the
    guards switch the culling off (distances beyond 1e12 and below 1e-6)."""
    def check(sc, W=64, H=36, samples=2, depth=6, seed=3):
        want_rgb, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
        rt.upload_scene(sc)
        rgb, em, _ = rt.render(W, H, samples, depth, seed, emission=True)
        assert np.array_equal(rgb, want_rgb), sc.name
        assert np.array_equal(em.view(np.uint32), want_em.view(np.uint32)), sc.name

    sc = scenes.many_objects(n=300, n_lights=0, seed=17)
    sc.dirlight((0.0, -1.0, 0.0))
    sc.dirlight((1.0, -1.0, 0.0))
    sc.dirlight((0.0, -1.0, -1.0))
    sc.name = "axis_lights"
    check(sc)

    sc = scenes.many_objects(n=300, n_lights=3, seed=19)
    sc.plane((0.0, -0.25, 0.0), (0.0, 1.0, 0.0), 1)
    sc.plane((0.0, 0.0, -70.0), (0.0, 0.0, 1.0), 3)
    sc.cube((2.0, 1.5, -6.0), (-1.0, -1.0, -1.0), 2)              # corners the other way round
    sc.cube((-3.0, 0.0, -5.0), (0.0, 1.0, 1.0), 1)                # flat
    sc.sphere((0.5, 0.5, -4.0), 0.0, 2)
    sc.sphere((-1.5, 0.6, -4.0), -0.6, 3)                         # sphere.h:19 squares the radius
    sc.sphere((3.0, 0.5, -8.0), float("inf"), 1)
    sc.cube((float("nan"), 0.0, -9.0), (1.0, 1.0, 1.0), 2)
    sc.name = "planes_and_degenerates"
    check(sc)

    base = scenes.many_objects(n=300, n_lights=4, seed=29)
    for k in (1e5, 1e9, 1e11, 1e14, 1e-4, 1e-7):
        check(_scaled_about_camera(base, k))


def test_cluster_skip_rule_never_skips_a_hit(rt):
    """The skip rule of the large-scene kernels on its own (rt_debug_cull) against the primitive tests on their own
    (rt_debug_intersect), for rays aimed to graze: impact parameters within 1e-7 .. 1e-1 of a sphere's radius, lines
    through a cube's corners, edges and faces offset by as little, origins from inside the hittable to 1e6 sizes away
    and behind it, directions with zero components and lengths other than 1, hittables from 1e-3 to 1e4 in size and at
    large coordinates, components of 1e-30 .. 1e-10 (a ray refracted into a glass cube comes out of refract() like that).
    A ray the rule would skip must never hit; and the rule must skip most of the rays that miss.  The distance rule
    (rt_debug_cull_enter) likewise: a hit at distance t lies in a box the ray enters no later than max(1.001 t, 0), so a
    closest-hit query that has a best distance b never drops a cluster holding a hit below b."""
    rs = np.random.RandomState(12345)
    eps = np.array([0.0, 1e-7, 3e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1])
    back = np.array([-3.0, -1.01, -0.5, 0.0, 0.5, 1.01, 2.0, 10.0, 100.0, 1e4, 1e6])

    def directions(n):
        d = rs.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        axis = rs.randint(0, 4, size=n)                      # a quarter each: one, two zero components
        for k in range(n):
            if axis[k] == 1:
                d[k, rs.randint(3)] = 0.0
            elif axis[k] == 2:
                z = rs.choice(3, 2, replace=False); d[k, z] = 0.0
            elif axis[k] == 3 and k % 2:                      # ... and of those left, half: one or two tiny components
                z = rs.choice(3, rs.randint(1, 3), replace=False)
                d[k, z] = rs.choice([-1.0, 1.0], size=len(z)) * 10.0 ** rs.uniform(-30, -10, size=len(z))
            if not d[k].any():
                d[k, 0] = 1.0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return d * rs.choice([1.0, 1.0, 0.5, 3.0, 1e-2, 50.0], size=(n, 1))

    def perpendicular(d):
        u = np.cross(d, rs.normal(size=d.shape))
        n = np.linalg.norm(u, axis=1, keepdims=True)
        u = np.where(n > 0, u / np.where(n > 0, n, 1), 0)
        return u

    n = 40000
    skipped_misses = misses = ruled = 0
    cases = [("sphere", (0.0, 0.0, -5.0), 1.0), ("sphere", (3.0, -2.0, 7.0), 1e-3), ("sphere", (0.0, -1000.0, -4.0), 1000.0),
             ("sphere", (1e5, 2e5, -3e5), 10.0), ("sphere", (0.5, 0.5, 0.5), 1e4),
             ("cube", (-1.0, 0.0, -6.0), (2.0, 1.0, 0.5)), ("cube", (5.0, 5.0, 5.0), (1e-3, 2e-3, 1e-3)),
             ("cube", (2.0, 1.5, -6.0), (-1.0, -1.0, -1.0)), ("cube", (-3e4, 1e4, 2e4), (100.0, 300.0, 50.0)),
             ("cube", (0.0, 0.0, 0.0), (1.0, 0.0, 1.0))]
    for kind, a, b in cases:
        sc = scenes.Scene(name="graze")
        sc.material((0.5, 0.5, 0.5), 0.0, 0.0, 0.0)
        sc.dirlight((0.0, -1.0, 0.0))
        d = directions(n)
        dn = d / np.linalg.norm(d, axis=1, keepdims=True)
        e = rs.choice(eps, size=n) * rs.choice([-1.0, 1.0], size=n)
        t0 = rs.choice(back, size=n)
        if kind == "sphere":
            sc.sphere(a, b, 0)
            c, r = np.array(a), float(b)
            target = c + perpendicular(dn) * (r * (1.0 + e))[:, None]
            size = r
        else:
            sc.cube(a, b, 0)
            lo, hi = np.minimum(np.array(a), np.array(a) + np.array(b)), np.maximum(np.array(a), np.array(a) + np.array(b))
            size = float(np.linalg.norm(hi - lo))
            # a point of the surface: every coordinate at lo, hi or in between (corners, edges, faces), pushed off by e
            sel = rs.randint(0, 3, size=(n, 3))
            target = np.where(sel == 0, lo, np.where(sel == 1, hi, lo + rs.uniform(size=(n, 3)) * (hi - lo)))
            target = target + perpendicular(dn) * (size * e)[:, None]
        origin = target - dn * (t0 * size)[:, None]
        rays = np.concatenate([origin, d], axis=1).astype(np.float32)
        # ... and rays that have nothing to do with the hittable
        far = np.concatenate([rs.normal(size=(n // 4, 3)) * size * 10 + np.array(a), directions(n // 4)], axis=1).astype(np.float32)
        rays = np.concatenate([rays, far])
        rt.upload_scene(sc)
        hit = rt.debug_intersect(0, rays)[:, 0] != 0
        may = rt.debug_cull(0, rays) != 0
        assert not (hit & ~may).any(), (kind, a, b, rays[hit & ~may][:3])
        res = rt.debug_intersect(0, rays)
        enter = rt.debug_cull_enter(0, rays)
        late = hit & (enter > np.maximum(np.float32(1.001) * res[:, 1], np.float32(0.0)))
        assert not late.any(), (kind, a, b, rays[late][:3], res[late][:3, 1], enter[late][:3])
        ruled += int((hit & (enter > -3e38)).sum())
        misses += int((~hit).sum())
        skipped_misses += int((~hit & ~may).sum())
    print("skip rule: %d of %d missing rays skipped; distance rule checked on %d hits" % (skipped_misses, misses, ruled))
    assert skipped_misses > 0.3 * misses
    assert ruled > 10000


def test_large_scene_more_pixels_than_resident_waves(rt):
    """coop_kernel's grid is only as large as fits the GPU at once; its waves take pixels from a queue until the list is
    used up.  A frame with several times more bounce pixels than resident waves (16 384 at most: 4 workgroups of 8
    waves on each of 256 CUs, times two for headroom) makes every wave render many pixels."""
    sc = scenes.many_objects(n=300, n_lights=2, seed=23)
    W, H, samples, depth, seed = 800, 450, 1, 3, 6
    want_rgb, _, _ = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8, want_emission=False)
    rt.upload_scene(sc)
    rgb, _, _ = rt.render(W, H, samples, depth, seed)
    assert np.array_equal(rgb, want_rgb)


def test_large_scene_ties_across_types_go_to_the_lower_scene_index(rt):
    """coop_kernel stages the hittables sorted by type, but equal distances must still go to the hittable that comes
    first in the scene (hittablesList.h:25-27).  A cube early in the scene whose front face lies in a plane that comes
    later (and the other way round), different materials, seen directly and through mirrors: plane and cube compute
    the distance differently, and for a good share of the rays the two floats are equal."""
    for first in ("cube", "plane"):
        sc = scenes.Scene(name="ties_" + first)
        scenes._default_materials(sc)
        mirror = sc.material((0.9, 0.9, 0.9), 0.0, 0.9, 0.0)
        rough = sc.material((0.2, 0.9, 0.3), 0.3, 0.6, 0.0)
        sc.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
        def cube(): sc.cube((-4.0, 0.0, -9.0), (8.0, 5.0, 2.0), mirror)            # front face z = -7
        def plane(): sc.plane((0.0, 0.0, -7.0), (0.0, 0.0, 1.0), rough)
        (cube if first == "cube" else plane)()
        sc.sphere((-2.0, 1.0, -2.0), 1.0, mirror)
        sc.sphere((2.5, 1.2, -3.0), 1.2, 3)
        rs = np.random.RandomState(5)
        for k in range(270):                                                       # >= 256 hittables: cooperative route
            x, z, r = rs.uniform(-30, 30), rs.uniform(-60, -12), rs.uniform(0.2, 0.6)
            if k % 3 == 2:
                sc.cube((x, 0.0, z), (r, r, r), k % 4)
            else:
                sc.sphere((x, r, z), r, k % 4)
        (plane if first == "cube" else cube)()
        sc.dirlight((-0.25, -0.75, 0.45))
        sc.pointlight((6.0, 9.0, 2.0))
        W, H, samples, depth, seed = 160, 90, 3, 6, 2
        want_rgb, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
        rt.upload_scene(sc)
        rgb, em, _ = rt.render(W, H, samples, depth, seed, emission=True)
        assert np.array_equal(rgb, want_rgb), first
        assert np.array_equal(em.view(np.uint32), want_em.view(np.uint32)), first


def test_capacity_and_sample_count_edges(rt):
    """RT_MAX_HITTABLES hittables: the staged scene leaves no LDS for the generators, so the listed kernels keep them in
    scratch and neither the sample-parallel nor the cooperative route applies.  More than 64 samples per pixel (one
    wave's lanes) and fewer than 8 also keep every pixel serial; 8 and 64 are the ends of the sample-parallel range."""
    from raytracer_amd import _abi
    sc = scenes.many_objects(n=4095, n_lights=2, seed=3)
    assert len(sc.hittables) == 4096
    want, want_em, _ = oracle_lib.render(sc, 48, 24, 2, 4, 7, threads=8)
    rt.upload_scene(sc)
    rgb, em, _ = rt.render(48, 24, 2, 4, 7, emission=True)
    assert np.array_equal(rgb, want) and np.array_equal(em.view(np.uint32), want_em.view(np.uint32))

    sc = scenes.glass_mirror()
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, 64, 36)
    for samples in (70, 7, 8, 64):
        want, _, _ = oracle_lib.render(sc, 64, 36, samples, 8, 2, threads=8)
        for flags in (0, _abi.RT_FLAG_FORCE_SAMPLE_PARALLEL):
            got = np.zeros((36, 64, 3), np.uint8)
            rt.render_rows(cam, rt.params(64, 36, samples, 8, 2, flags=flags), got)
            assert np.array_equal(got, want), (samples, flags)


def test_tile_stream_matches_reference_records(rt):
    """rt_render_tiles hands out what the reference pushes to its Redis list per 16x16 tile (src/main.cpp:322-326,
    src/redis.h:27-36): [int x][int y][pixels] then [int x][int y][emissionPixels].  Reassembled like the reference's
    consumer does (src/main.cpp:378-395), the records give the oracle's frame; partial tiles at the right / bottom edge
    are not produced (src/main.cpp:276)."""
    sc = scenes.glass_mirror()
    W, H, samples, depth, seed, T = 150, 70, 2, 6, 3, 16
    want, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
    rt.upload_scene(sc)
    for rows_per_chunk in (0, 16, 40):
        recs = rt.render_tiles(W, H, samples, depth, seed, tile=T, rows_per_chunk=rows_per_chunk)
        tx, ty = W // T, H // T
        assert len(recs) == 2 * tx * ty
        img = np.zeros((ty * T, tx * T, 3), np.uint8)
        em = np.zeros((ty * T, tx * T, 4), np.float32)
        for k in range(0, len(recs), 2):
            a, b = recs[k], recs[k + 1]
            assert len(a) == 8 + T * T * 3 and len(b) == 8 + T * T * 16
            x, y = np.frombuffer(a[:8], np.int32)
            assert tuple(np.frombuffer(b[:8], np.int32)) == (x, y) and (x, y) == ((k // 2) % tx, (k // 2) // tx)
            img[y * T:(y + 1) * T, x * T:(x + 1) * T] = np.frombuffer(a[8:], np.uint8).reshape(T, T, 3)
            em[y * T:(y + 1) * T, x * T:(x + 1) * T] = np.frombuffer(b[8:], np.float32).reshape(T, T, 4)
        assert np.array_equal(img, want[:ty * T, :tx * T])
        assert np.array_equal(em.view(np.uint32), want_em[:ty * T, :tx * T].view(np.uint32))
    # an odd tile size, and a sink that stops the stream
    recs = rt.render_tiles(W, H, samples, depth, seed, tile=5)
    assert len(recs) == 2 * (W // 5) * (H // 5) and np.array_equal(
        np.frombuffer(recs[2][8:], np.uint8).reshape(5, 5, 3), want[0:5, 5:10])
    seen = []
    with pytest.raises(ra._abi.RtError):
        rt.render_tiles(W, H, samples, depth, seed, tile=T, sink=lambda rec: seen.append(rec) or len(seen) >= 3)
    assert len(seen) == 3


def test_executed_tally_prunes_only_shadow_queries(rt):
    """RT_FLAG_COUNT_EXECUTED tallies the production kernel: same image, same closest-hit / draw counts, fewer (or
    equal) shadow queries and primitive tests than the reference issues."""
    sc = scenes.point_light_scene()
    W, H, samples, depth, seed = 96, 64, 2, 6, 5
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    p = rt.params(W, H, samples, depth, seed)
    a = np.zeros((H, W, 3), np.uint8); b = np.zeros((H, W, 3), np.uint8)
    ref = rt.render_rows(cam, p, a, count=True)
    exe = rt.render_rows(cam, p, b, count=True, executed=True)
    assert np.array_equal(a, b)
    for k in ("primary", "closest", "rand_draws", "shaded"):
        assert ref[k] == exe[k], k
    assert exe["shadow"] < ref["shadow"] and exe["tests"] < ref["tests"]
    assert ref["shadow"] == ref["shaded"] * len(sc.lights)


def test_black_materials_shade_without_their_shadow_queries(rt):
    """A material whose colour is zero in every channel shades to (0 * intensity) * multiplier whatever the lights say, so
    the production kernels leave its shadow queries out (trace_path).  Same bytes and bits as the reference's, which asks
    them: black mirrors (rough and perfect), black glass, a black reflect-and-refract material, channels of -0.0 (the
    product keeps the zero's sign), a colour of 1e-30 as the control that must NOT take the shortcut; point lights and a
    directional one; seen directly, in mirrors and through glass; small scene (listed kernels) and large (cooperative).
    The tallying builds still ask and count every query the reference asks."""
    for big in (False, True):
        sc = scenes.Scene(name="black_big" if big else "black")
        scenes._default_materials(sc)                                      # 3 = the reference's glass: colour (0, 0, 0)
        bm = sc.material((0.0, 0.0, 0.0), 0.2, 0.8, 0.0)                   # black rough mirror
        bp = sc.material((-0.0, 0.0, -0.0), 0.0, 0.9, 0.0)                 # perfect mirror, negative zeros
        bx = sc.material((0.0, -0.0, 0.0), 0.05, 0.5, 1.5, (0.2, 0.9, 0.1), 2.0)    # reflects and refracts, glows
        nb = sc.material((1e-30, 0.0, 0.0), 0.1, 0.7, 0.0)                 # not black
        sc.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
        sc.sphere((0.0, 1.0, -4.0), 1.0, bm)
        sc.sphere((-2.5, 1.0, -3.0), 1.0, bp)
        sc.sphere((2.5, 1.0, -3.0), 1.0, bx)
        sc.sphere((-1.2, 0.6, -1.0), 0.6, 3)
        sc.cube((1.0, 0.0, -1.5), (0.8, 0.8, 0.8), 3)
        sc.sphere((4.5, 1.2, -6.0), 1.2, nb)
        sc.cube((-4.5, 0.0, -7.0), (1.5, 2.0, 1.5), bm)
        sc.dirlight((-0.25, -0.75, 0.45))
        sc.pointlight((2.0, 6.0, 1.0))
        sc.pointlight((-5.0, 3.0, -2.0))
        if big:
            rs = np.random.RandomState(11)
            for k in range(280):
                x, z, r = rs.uniform(-30, 30), rs.uniform(-60, -10), rs.uniform(0.2, 0.6)
                m = (0, 1, 2, 3, bm, bp, bx, nb)[k % 8]
                if k % 3 == 2:
                    sc.cube((x, 0.0, z), (r, r, r), m)
                else:
                    sc.sphere((x, r, z), r, m)
        W, H, samples, depth, seed = (160, 90, 2, 8, 3) if big else (200, 120, 6, 12, 3)
        want_rgb, want_em, want_cnt = oracle_lib.render(sc, W, H, samples, depth, seed, threads=8)
        rt.upload_scene(sc)
        rgb, em, _ = rt.render(W, H, samples, depth, seed, emission=True)
        assert np.array_equal(rgb, want_rgb), sc.name
        assert np.array_equal(em.view(np.uint32), want_em.view(np.uint32)), sc.name
        cam = ra.camera35f(sc, W, H)
        p = rt.params(W, H, samples, depth, seed)
        a = np.zeros((H, W, 3), np.uint8)
        ref = rt.render_rows(cam, p, a, count=True)
        assert np.array_equal(a, want_rgb)
        for k in ("closest", "shadow", "tests", "rand_draws"):
            assert ref[k] == want_cnt[k], (sc.name, k)


def test_kernel_pair_equals_single_general_kernel(rt):
    """Default path = classify + direct kernel + listed general kernel; RT_FLAG_SINGLE_KERNEL = the general kernel over
    every pixel.  Same bytes; checked against the oracle on a band."""
    from raytracer_amd import _abi
    for sc, depth in ((scenes.glass_mirror(), 12), (scenes.default4(), 0), (scenes.cube_scene(), 5)):
        W, H, samples, seed = 640, 360, 4, 3
        rt.upload_scene(sc)
        cam = ra.camera35f(sc, W, H)
        a = np.zeros((H, W, 3), np.uint8); b = np.zeros((H, W, 3), np.uint8)
        ea = np.zeros((H, W, 4), np.float32); eb = np.zeros((H, W, 4), np.float32)
        rt.render_rows(cam, rt.params(W, H, samples, depth, seed), a, ea)
        t_pair = rt.last_device_ms
        rt.render_rows(cam, rt.params(W, H, samples, depth, seed, flags=_abi.RT_FLAG_SINGLE_KERNEL), b, eb)
        print("%s: device ms kernel pair %.2f, single kernel %.2f" % (sc.name, t_pair, rt.last_device_ms))
        assert np.array_equal(a, b) and np.array_equal(ea.view(np.uint32), eb.view(np.uint32))
        want, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, seed, row0=150, rows=48, threads=8)
        assert np.array_equal(a[150:198], want) and np.array_equal(ea[150:198].view(np.uint32), want_em.view(np.uint32))


def test_sample_parallel_route_equals_serial(rt):
    """Long pixels are finished one lane per sample from guessed generator states (heavy_kernel).  The test hook sends
    every bounce pixel that way: rough materials make the guesses fail (re-guess path), glass makes them hold.  Same
    bytes as the serial general kernel, and as the oracle on a band."""
    from raytracer_amd import _abi
    cases = ((scenes.glass_mirror(), 12, 7), (scenes.default4(), 10, 64), (scenes.random10(), 6, 16),
             (scenes.cube_scene(), 5, 2), (scenes.point_light_scene(), 4, 33))
    for sc, depth, samples in cases:
        W, H, seed = 320, 180, 5
        rt.upload_scene(sc)
        cam = ra.camera35f(sc, W, H)
        a = np.zeros((H, W, 3), np.uint8); b = np.zeros((H, W, 3), np.uint8)
        ea = np.zeros((H, W, 4), np.float32); eb = np.zeros((H, W, 4), np.float32)
        rt.render_rows(cam, rt.params(W, H, samples, depth, seed, flags=_abi.RT_FLAG_FORCE_SAMPLE_PARALLEL), a, ea)
        rt.render_rows(cam, rt.params(W, H, samples, depth, seed, flags=_abi.RT_FLAG_SINGLE_KERNEL), b, eb)
        assert np.array_equal(a, b), sc.name
        assert np.array_equal(ea.view(np.uint32), eb.view(np.uint32)), sc.name
        # without the emission buffer another kernel instance runs
        c = np.zeros((H, W, 3), np.uint8)
        rt.render_rows(cam, rt.params(W, H, samples, depth, seed, flags=_abi.RT_FLAG_FORCE_SAMPLE_PARALLEL), c, None)
        assert np.array_equal(c, b), sc.name
    sc = scenes.default4()
    rt.upload_scene(sc)
    W, H, samples, depth, seed = 320, 180, 64, 10, 5
    want, _, _ = oracle_lib.render(sc, W, H, samples, depth, seed, row0=100, rows=24, threads=8)
    got, _, _ = rt.render(W, H, samples, depth, seed, row0=100, rows=24)          # default route, long pixels handed over
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,W,H,samples,depth", [
    ("default4", 333, 77, 8, 10),
    ("default4", 160, 90, 1, 10),            # four rays per pixel: one lane in four of a group has a ray
    ("default4", 96, 54, 67, 10),            # more rays than a 64-sample pixel
    ("glass_mirror", 200, 120, 16, 12),      # mixed material: draw counts change from ray to ray
    ("glass_mirror", 64, 64, 5, 20),         # deep stack variant
    ("random10", 256, 128, 9, 12),
    ("cube", 96, 64, 6, 5),
    ("degenerate", 120, 72, 4, 64),
])
def test_listed_routes_give_the_same_bytes(rt, name, W, H, samples, depth):
    """The bounce pixels' routes -- a frame on its own (two listed launches side by side, long steady pixels handed to the
    sample-parallel kernels), a frame of a sequence (one launch, generators in scratch, the direct kernel on a shared
    stream) and every bounce pixel forced down the sample-parallel route -- with the listed kernel's per-wave candidate
    lists: same bytes as the oracle, with and without the emission buffer.  (The cases are round 2's for its ray-parallel
    route, which round 3 removed.)"""
    from raytracer_amd import _abi
    sc = scenes.BUILTIN[name]()
    want_rgb, want_em, _ = oracle_lib.render(sc, W, H, samples, depth, 31, threads=8)
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    for flags in (0, _abi.RT_FLAG_SEQUENCE, _abi.RT_FLAG_FORCE_SAMPLE_PARALLEL):
        rgb = np.zeros((H, W, 3), np.uint8); em = np.zeros((H, W, 4), np.float32)
        rt.render_rows(cam, rt.params(W, H, samples, depth, 31, flags=flags), rgb, em)
        assert np.array_equal(rgb, want_rgb), flags
        assert np.array_equal(em.view(np.uint32), want_em.view(np.uint32)), flags
        rgb = np.zeros((H, W, 3), np.uint8)
        rt.render_rows(cam, rt.params(W, H, samples, depth, 31, flags=flags), rgb, None)
        assert np.array_equal(rgb, want_rgb), (flags, "rgb only")


def test_rows_and_cyclic_bands_are_partition_independent(rt):
    sc = scenes.glass_mirror()
    W, H, samples, depth, seed = 120, 100, 2, 6, 9
    rt.upload_scene(sc)
    full, full_em, _ = rt.render(W, H, samples, depth, seed, emission=True)
    # contiguous slabs
    top, _, _ = rt.render(W, H, samples, depth, seed, row0=0, rows=37)
    bot, _, _ = rt.render(W, H, samples, depth, seed, row0=37, rows=H - 37)
    assert np.array_equal(np.concatenate([top, bot]), full)
    # cyclic bands of 8 rows over 3 ranks (H = 100 is not a multiple of 24: the last turn is ragged)
    band, nranks = 8, 3
    frame = np.zeros_like(full)
    for rank in range(nranks):
        ys = [y for y in range(H) if (y // band) % nranks == rank]
        rgb, _, _ = rt.render(W, H, samples, depth, seed, row0=rank * band, rows=len(ys), band_rows=band, band_stride=band * nranks)
        frame[ys] = rgb
    assert np.array_equal(frame, full)


def test_device_output_buffers_and_deinterleave(rt):
    import torch
    sc = scenes.default4()
    W, H, samples, depth, seed = 128, 72, 2, 10, 0
    rt.upload_scene(sc)
    full, full_em, _ = rt.render(W, H, samples, depth, seed, emission=True)
    band, nranks = 8, 2
    nbands = (H + band - 1) // band
    rows_per_rank = ((nbands + nranks - 1) // nranks) * band
    gathered = torch.zeros((nranks, rows_per_rank, W, 3), dtype=torch.uint8, device="cuda:0")
    gathered_em = torch.zeros((nranks, rows_per_rank, W, 4), dtype=torch.float32, device="cuda:0")
    cam = ra.camera35f(sc, W, H)
    for rank in range(nranks):
        nrows = sum(1 for y in range(H) if (y // band) % nranks == rank)
        p = rt.params(W, H, samples, depth, seed, row0=rank * band, rows=nrows, band_rows=band, band_stride=band * nranks)
        rt.render_rows(cam, p, gathered[rank], gathered_em[rank])
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    frame_em = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    rt.deinterleave_bands(gathered, frame, W, H, 3, band, nranks, rows_per_rank)
    rt.deinterleave_bands(gathered_em, frame_em, W, H, 16, band, nranks, rows_per_rank)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), full)
    assert np.array_equal(frame_em.cpu().numpy().view(np.uint32), full_em.view(np.uint32))


@pytest.mark.parametrize("name,W,H,samples,depth", [("default4", 333, 77, 8, 10), ("glass_mirror", 200, 120, 16, 12), ("cube", 96, 64, 3, 5)])
def test_sequence_hint_gives_the_same_bytes(rt, name, W, H, samples, depth):
    """RT_FLAG_SEQUENCE changes how the frame is scheduled (listed kernels' generators in scratch, one listed launch,
    the direct kernel on a shared stream, host waits that poll with naps), not its bytes -- and the polling leaves no
    hipErrorNotReady behind as the thread's last HIP error (torch checks that after its next kernel launch)."""
    import torch
    from raytracer_amd import _abi
    sc = scenes.BUILTIN[name]()
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    out = []
    for flags in (0, _abi.RT_FLAG_SEQUENCE, _abi.RT_FLAG_SEQUENCE | _abi.RT_FLAG_FORCE_SAMPLE_PARALLEL):
        rgb = np.zeros((H, W, 3), np.uint8); em = np.zeros((H, W, 4), np.float32)
        rt.render_rows(cam, rt.params(W, H, samples, depth, 11, flags=flags), rgb, em)
        assert int(torch.ones(8, device="cuda:0").sum().item()) == 8
        out.append((rgb, em))
    for rgb, em in out[1:]:
        assert np.array_equal(rgb, out[0][0]) and np.array_equal(em.view(np.uint32), out[0][1].view(np.uint32))


@pytest.mark.parametrize("name,W,H,samples,depth", [("default4", 333, 77, 8, 10), ("glass_mirror", 200, 120, 16, 12), ("cube", 96, 64, 3, 5),
                                                  ("plane", 130, 50, 5, 6), ("default4", 64, 64, 64, 3)])
def test_wave_per_workgroup_listed_kernel_gives_the_same_bytes(rt, name, W, H, samples, depth):
    """Frames in flight whose listed kernel is long launch it with one wave per workgroup (render_kernel<..., WG = 64>:
    a context chooses that from what its last frame measured; RT_FLAG_FORCE_WAVE_GROUPS is the test hook).  A launch
    shape, not arithmetic: the bytes, the emission bits and -- on the frame the oracle renders here -- the reference's
    pixels are the same, with and without an emission buffer, and the frames after it on the same context as well
    (the context's own choice, whatever it measured)."""
    from raytracer_amd import _abi
    sc = scenes.BUILTIN[name]()
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    ref = np.zeros((H, W, 3), np.uint8); ref_em = np.zeros((H, W, 4), np.float32)
    rt.render_rows(cam, rt.params(W, H, samples, depth, 23), ref, ref_em)
    if name == "cube":
        want = oracle_lib.render(sc, W, H, samples, depth, 23)[0]
        assert np.array_equal(ref, want)
    forced = _abi.RT_FLAG_SEQUENCE | _abi.RT_FLAG_FORCE_WAVE_GROUPS
    for flags in (forced, forced, _abi.RT_FLAG_SEQUENCE):
        rgb = np.zeros((H, W, 3), np.uint8); em = np.zeros((H, W, 4), np.float32)
        rt.render_rows(cam, rt.params(W, H, samples, depth, 23, flags=flags), rgb, em)
        assert np.array_equal(rgb, ref) and np.array_equal(em.view(np.uint32), ref_em.view(np.uint32))
        rgb2 = np.zeros((H, W, 3), np.uint8)
        rt.render_rows(cam, rt.params(W, H, samples, depth, 23, flags=flags), rgb2)
        assert np.array_equal(rgb2, ref)
    # together with the sample-parallel hook (that kernel then hands pixels over), and outside a sequence (ignored)
    for flags in (forced | _abi.RT_FLAG_FORCE_SAMPLE_PARALLEL, _abi.RT_FLAG_FORCE_WAVE_GROUPS):
        rgb = np.zeros((H, W, 3), np.uint8)
        rt.render_rows(cam, rt.params(W, H, samples, depth, 23, flags=flags), rgb)
        assert np.array_equal(rgb, ref)


def test_frames_in_flight_equal_frames_one_at_a_time(rt):
    """raytracer_amd/pipeline.py: several contexts rendering different frames at once on one GPU, from host threads,
    produce the bytes of the same frames rendered alone (glass: listed, sample-parallel and resume kernels overlap
    across contexts)."""
    import torch
    from raytracer_amd import _abi, bands
    from raytracer_amd.pipeline import FramePipeline
    sc = scenes.glass_mirror()
    W, H, samples, depth = 160, 96, 16, 12
    band, in_flight, n_frames = 16, 3, 7
    rt.upload_scene(sc)
    want = [rt.render(W, H, samples, depth, seed_base=1000 * k)[0] for k in range(n_frames)]
    cam = ra.camera35f(sc, W, H)
    rts = [ra.Renderer(0) for _ in range(in_flight)]
    try:
        for r in rts:
            r.upload_scene(sc)
        blocks = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(2 * in_flight)]
        torch.cuda.synchronize()

        def render(worker, block, k):
            p = bands.band_params(rts[worker], W, H, samples, depth, 1000 * k, band, 1, 0, flags=_abi.RT_FLAG_SEQUENCE if k % 2 else 0)
            rts[worker].render_rows(cam, p, block)

        got = {}
        FramePipeline(render, blocks, H, band, rts[0], workers=in_flight).run(n_frames, on_frame=lambda k, f: got.__setitem__(k, f.cpu().numpy()))
    finally:
        for r in rts:
            r.close()
    assert sorted(got) == list(range(n_frames))
    for k in range(n_frames):
        assert np.array_equal(got[k], want[k]), "frame %d" % k


def test_shared_direct_streams_come_and_go_with_the_contexts():
    """Under RT_FLAG_SEQUENCE the direct kernels of a device's contexts go round streams the contexts share
    (rt_abi.cpp, RT_SEQ_DIRECT_STREAMS); the last context of the device destroys them and the next one starts afresh.
    A process of its own, so that no other context keeps them alive: two contexts render in turn with the flag, are
    closed, two new ones render again -- always the bytes of a frame rendered without the flag."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import raytracer_amd as ra
        from raytracer_amd import _abi, bands, scenes
        sc = scenes.default4()
        W, H, samples, depth = 192, 96, 8, 6      # (8 samples: the fewest with which the flag changes the route)
        cam = ra.camera35f(sc, W, H)
        want = None
        for generation in range(3):
            rts = [ra.Renderer(0) for _ in range(2)]
            for r in rts:
                r.upload_scene(sc)
            if want is None:
                want = [rts[0].render(W, H, samples, depth, seed_base=5 * k)[0] for k in range(4)]
            for k in range(4):
                r = rts[k %% 2]
                p = bands.band_params(r, W, H, samples, depth, 5 * k, 16, 1, 0, flags=_abi.RT_FLAG_SEQUENCE)
                out = np.zeros((H, W, 3), np.uint8)
                r.render_rows(cam, p, out)
                assert np.array_equal(out, want[k]), (generation, k)
            for r in rts:
                r.close()
        print("ok")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_gather_rows_over_rccl_single_rank(rt):
    """rt_gather_rows with a real RCCL communicator of one rank: the rank's block goes through ncclSend / ncclRecv (to
    and from itself) into the gather buffer and is reassembled; same bytes as the frame rendered whole.  (More ranks
    need more GPUs: tests/test_bands_gloo.py covers the partition and the frame order with gloo.)"""
    import torch
    from raytracer_amd import _abi, bands
    sc = scenes.default4()
    W, H, samples, depth, band = 200, 90, 2, 10, 16
    rt.upload_scene(sc)
    want, _, _ = rt.render(W, H, samples, depth, 5)
    cam = ra.camera35f(sc, W, H)
    rpr = bands.rows_per_rank(H, band, 1)
    block = torch.zeros((rpr, W, 3), dtype=torch.uint8, device="cuda:0")
    gathered = torch.zeros((1, rpr, W, 3), dtype=torch.uint8, device="cuda:0")
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    rt.render_rows(cam, bands.band_params(rt, W, H, samples, depth, 5, band, 1, 0), block)
    comm = rt.comm_create(ra.Renderer.comm_unique_id(), 0, 1)
    try:
        for _ in range(2):
            frame.zero_()
            rt.gather_rows(comm, 0, 1, block, gathered, frame, W, H, 3, band, rpr)
            assert np.array_equal(frame.cpu().numpy(), want)
        # what bench.py --gpus N reports about the transport: the ranks RCCL itself counts, the gather's device time
        assert ra.Renderer.comm_count(comm) == 1
        assert 0.0 < rt.last_gather_ms < 1000.0
        # no communicator, one rank: reassembled straight from the block
        frame.zero_()
        rt.gather_rows(None, 0, 1, block, None, frame, W, H, 3, band, rpr)
        assert np.array_equal(frame.cpu().numpy(), want)
        with pytest.raises(ra.RtError) as e:                     # two ranks need a communicator
            rt.gather_rows(None, 0, 2, block, gathered, frame, W, H, 3, band, rpr)
        assert e.value.status == _abi.RT_ERR_INVALID_ARG
        with pytest.raises(ra.RtError) as e:                     # blocks too small for the frame (a short last band of
            rt.gather_rows(comm, 0, 2, block, gathered, frame, W, 101, 3, 100, 1)     # one rank, a full band of the other)
        assert e.value.status == _abi.RT_ERR_INVALID_ARG
        with pytest.raises(ra.RtError) as e:
            rt.deinterleave_bands(gathered, frame, W, 101, 3, 100, 2, 1)
        assert e.value.status == _abi.RT_ERR_INVALID_ARG
    finally:
        rt.comm_destroy(comm)


@pytest.mark.parametrize("name,W,H,samples", [("default4", 256, 128, 2), ("default4", 1024, 512, 1), ("glass_mirror", 192, 96, 2), ("default4", 48, 32, 1)])
def test_glow_matches_the_cpu_restatement(rt, name, W, H, samples):
    """f1: rt_apply_glow against oracle/rt_glow.c (applyGlow of src/main.cpp:163-229; the oracle is a restatement,
    PARITY UNPINNED: the reference's main.cpp does not build here).  Bytes identical, same number of pyramid levels; host
    and device buffers; the emission buffer is left as it was."""
    import torch
    sc = scenes.BUILTIN[name]()
    rt.upload_scene(sc)
    rgb, em, _ = rt.render(W, H, samples, 10, 3, emission=True)
    want, levels = oracle_lib.apply_glow(rgb, em)
    assert levels >= 1 and not np.array_equal(want, rgb)
    got, em_in = rgb.copy(), em.copy()
    assert rt.apply_glow(got, em_in, W, H) == levels
    assert np.array_equal(got, want) and np.array_equal(em_in.view(np.uint32), em.view(np.uint32))
    d_rgb, d_em = torch.from_numpy(rgb).cuda(), torch.from_numpy(em).cuda()
    assert rt.apply_glow(d_rgb, d_em, W, H) == levels
    assert np.array_equal(d_rgb.cpu().numpy(), want) and np.array_equal(d_em.cpu().numpy().view(np.uint32), em.view(np.uint32))


def test_glow_levels_that_do_not_divide_the_frame_are_refused(rt):
    """The reference indexes a level's upscaled image with the frame's row length (src/main.cpp:191-198): with sizes that
    2^level does not divide it reads out of bounds.  Both sides stop there: the oracle returns -1, the library
    RT_ERR_UNSUPPORTED, and the levels before it are applied."""
    from raytracer_amd import _abi
    sc = scenes.default4()
    rt.upload_scene(sc)
    W, H = 200, 100                                     # 100 / 8 is not whole: the third level is refused
    rgb, em, _ = rt.render(W, H, 1, 10, 0, emission=True)
    _, levels = oracle_lib.apply_glow(rgb, em)
    assert levels == -1
    with pytest.raises(ra.RtError) as e:
        rt.apply_glow(rgb.copy(), em, W, H)
    assert e.value.status == _abi.RT_ERR_UNSUPPORTED


def test_probe_max_depth(rt, golden_dir):
    z = np.load(os.path.join(golden_dir, "probe_glass_mirror_128x64_d3_seed0.npz"))
    sc = scenes.Scene.from_text(bytes(z["scene"]).decode())
    W, H, _, depth, seed = (int(v) for v in z["params"])
    rt.upload_scene(sc)
    assert np.array_equal(rt.probe_max_depth(W, H, depth, seed), z["flags"])


def test_rng_reproduces_glibc(rt, golden_dir):
    import json
    with open(os.path.join(golden_dir, "glibc_rand.json")) as fh:
        known = json.load(fh)
    seeds = [int(s) for s in known]
    out = rt.debug_rand(seeds, 96)
    for i, s in enumerate(seeds):
        assert out[i].tolist() == known[str(s)], s
    rs = np.random.RandomState(3)
    seeds = rs.randint(0, 2**32, size=1000, dtype=np.uint64).astype(np.uint32)
    out = rt.debug_rand(seeds, 40)
    for i in range(0, 1000, 37):
        assert out[i].tolist() == oracle_lib.rand_sequence(int(seeds[i]), 40)


def test_short_sqrt_and_reciprocal_are_ieee_for_every_float(rt):
    """The kernels' sqrt / 1 / y / 1 / sqrt sequences (rt_device.hpp rt_sqrt, rt_rcp, rt_inv_sqrt) are shorter than the
    compiler's; the reference's are IEEE (sphere.h:25, glm normalize).  All 2^32 bit patterns, bit for bit."""
    assert rt.debug_exact_math() == (0, 0, 0)


def test_quantise_matches_double_sqrt(rt):
    rs = np.random.RandomState(5)
    vals = [rs.uniform(0, 1, 200000).astype(np.float32), np.linspace(0, 1, 65537, dtype=np.float32)]
    # floats on either side of every 8-bit boundary q: c = (q / 255)^2
    q = np.arange(0, 256, dtype=np.float64)
    edge = ((q / 255.0) ** 2).astype(np.float32)
    for k in range(-3, 4):
        e = edge.copy()
        for _ in range(abs(k)):
            e = np.nextafter(e, np.float32(2.0 if k > 0 else -1.0))
        vals.append(np.clip(e, 0, 1))
    vals.append(np.asarray([np.nan, 0.0, 1.0], dtype=np.float32))
    v = np.concatenate(vals)
    got = rt.debug_quantise(v)
    with np.errstate(invalid="ignore"):
        want = np.where(np.isnan(v), 0, (np.sqrt(v.astype(np.float64)) * 255.0)).astype(np.int64).astype(np.uint8)
    assert np.array_equal(got, want)


def test_primitive_known_answers(rt):
    s = scenes.Scene()
    s.material((1, 1, 1), 0, 0, 0)
    s.sphere((0, 0, -5), 1.0, 0)
    s.plane((0, 0, 0), (0, 1, 0), 0)
    s.cube((-1, -1, -6), (2, 2, 2), 0)
    rt.upload_scene(s)
    # sphere.h: unit-direction ray along -z from the origin hits the unit sphere at z=-5 at distance 4, normal +z
    o = rt.debug_intersect(0, [[0, 0, 0, 0, 0, -1], [0, 0, 0, 0, 0, 1], [0, 0, -5, 0, 0, -1], [0, 2, 0, 0, 0, -1]])
    assert o[0].tolist() == [1, 4, 0, 0, -4, 0, 0, 1]
    assert o[1][0] == 0                       # both roots negative
    assert o[2].tolist() == [1, 1, 0, 0, -6, 0, 0, -1]   # origin inside: far root, outward normal
    assert o[3][0] == 0                       # discriminant < 0
    # sphere.h:26-27 does not divide by a: with direction (0,0,-2) t = 8 (not 2) and the "hit" lands at z = -16
    o = rt.debug_intersect(0, [[0, 0, 0, 0, 0, -2]])
    assert o[0][0] == 1 and o[0][1] == 8.0 and o[0][4] == -16.0
    # plane.h:19 sign quirk: a ray going down towards y=0 from above gets t < 0 (miss); going up it "hits"
    o = rt.debug_intersect(1, [[0, 1, 0, 0, -1, 0], [0, 1, 0, 0, 1, 0], [0, 1, 0, 1, 0, 0]])
    assert o[0][0] == 0
    assert o[1].tolist() == [1, 1, 0, 2, 0, 0, 1, 0]
    assert o[2][0] == 0                       # |denom| <= 1e-6
    # cube.h: slab test, always tMin; axis-parallel rays rely on IEEE infinities
    o = rt.debug_intersect(2, [[0, 0, 0, 0, 0, -1], [0, 0, -5, 0, 0, -1], [5, 0, 0, 0, 0, -1]])
    assert o[0].tolist() == [1, 4, 0, 0, -4, 0, 0, 1]
    assert o[1][0] == 1 and o[1][1] == -1.0   # origin inside: negative tMin is returned as the distance
    assert o[2][0] == 0


def test_error_codes(rt):
    from raytracer_amd import _abi
    sc = scenes.default4()
    rt.upload_scene(sc)
    with pytest.raises(ra.RtError) as e:
        rt.render(64, 64, 1, 65)                       # max_depth > RT_MAX_DEPTH
    assert e.value.status == _abi.RT_ERR_UNSUPPORTED
    with pytest.raises(ra.RtError) as e:
        rt.render(64, 64, 0, 5)
    assert e.value.status == _abi.RT_ERR_INVALID_ARG
    with pytest.raises(ra.RtError) as e:
        rt.render(64, 64, 1, 5, row0=60, rows=10)      # rows outside the frame
    assert e.value.status == _abi.RT_ERR_INVALID_ARG
    bad = scenes.default4(); bad.hittables[1] = (0, (0.0, 1.0, -4.0), (0.0, 0.0, 0.0), 1.0, 9)
    with pytest.raises(ra.RtError) as e:
        rt.upload_scene(bad)
    assert e.value.status == _abi.RT_ERR_SCENE
    hot = scenes.default4(); hot.materials[2] = hot.materials[2][:4] + (1.5,) + hot.materials[2][5:]
    with pytest.raises(ra.RtError) as e:
        rt.upload_scene(hot)
    assert e.value.status == _abi.RT_ERR_UNSUPPORTED
    for col in (4, 5):                                # negative reflection / refraction: inf / NaN pixels in the reference
        neg = scenes.default4(); neg.materials[1] = neg.materials[1][:col] + (-0.25,) + neg.materials[1][col + 1:]
        with pytest.raises(ra.RtError) as e:
            rt.upload_scene(neg)
        assert e.value.status == _abi.RT_ERR_UNSUPPORTED
    fresh = ra.Renderer(0)
    with pytest.raises(ra.RtError) as e:
        fresh.render(64, 64, 1, 5, cam=ra.camera35f(60.0, 64, 64))
    assert e.value.status == _abi.RT_ERR_NO_SCENE
    fresh.close()
    # empty scene and zero rows are legal
    empty = scenes.Scene(); rt.upload_scene(empty)
    rgb, _, _ = rt.render(32, 16, 1, 5)
    want, _, _ = oracle_lib.render(empty, 32, 16, 1, 5)
    assert np.array_equal(rgb, want)
    rgb, _, _ = rt.render(32, 16, 1, 5, row0=3, rows=0)
    assert rgb.shape == (0, 32, 3)
