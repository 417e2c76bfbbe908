"""Pins the CPU restatement (oracle/rt_oracle.c) against fixtures produced by the REAL reference
(oracle/_ref/ref_oracle = unmodified reference headers; tests/golden/make_golden.py) and against glibc."""
import glob
import json
import os

import numpy as np
import pytest

import oracle_lib
from raytracer_amd import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith(("probe_", "fullsize")))


def load_case(path):
    z = np.load(path)
    sc = scenes.Scene.from_text(bytes(z["scene"]).decode())
    W, H, samples, depth, seed = (int(v) for v in z["params"])
    return z, sc, W, H, samples, depth, seed


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_golden(path):
    z, sc, W, H, samples, depth, seed = load_case(path)
    rgb, em, cnt = oracle_lib.render(sc, W, H, samples, depth, seed)
    assert np.array_equal(rgb, z["rgb"]), "8-bit image differs from the reference's"
    assert np.array_equal(em.view(np.uint32), z["emission"].view(np.uint32)), "emission buffer is not bit-identical"
    assert [cnt[k] for k in ("primary", "closest", "shadow", "tests", "rand_draws")] == [int(v) for v in z["counters"]]
    assert np.array_equal(oracle_lib.camera(sc, W, H).view(np.uint32), z["camera"].view(np.uint32))


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_oracle_matches_reference_at_full_size(name):
    """BASELINE.json configs C1 / C2 of DEFAULT-4: the port's whole frame against the hashes and tallies of the real
    reference's frame (tests/golden/make_fullsize.py; SURVEY.md Appendix C: 572aa677 / 10 710 164 and 82bdc0e9 / 135 900 201)."""
    import hashlib
    with open(os.path.join(GOLDEN, "fullsize_hashes.json")) as fh:
        g = json.load(fh)[name]
    assert g["fnv1a32_body"] == {"C1": "572aa677", "C2": "82bdc0e9"}[name]
    z = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    W, H, samples, depth, seed = g["width"], g["height"], g["samples"], g["max_depth"], g["seed_base"]
    rgb, em, cnt = oracle_lib.render(scenes.default4(), W, H, samples, depth, seed, threads=8)
    assert np.array_equal(rgb[::g["row_step"]], z[name + "_rows"])
    assert np.array_equal(em[::g["row_step"]].view(np.uint32), z[name + "_emission_rows"].view(np.uint32))
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == g["sha256_body"]
    assert {k: cnt[k] for k in g["counters"]} == g["counters"]


def test_oracle_is_thread_count_and_band_independent():
    z, sc, W, H, samples, depth, seed = load_case(os.path.join(GOLDEN, "glass_mirror_96x80_s2_d3_seed9.npz"))
    rgb4, em4, cnt4 = oracle_lib.render(sc, W, H, samples, depth, seed, threads=4)
    assert np.array_equal(rgb4, z["rgb"]) and np.array_equal(em4.view(np.uint32), z["emission"].view(np.uint32))
    assert cnt4["closest"] == int(z["counters"][1])
    top, _, _ = oracle_lib.render(sc, W, H, samples, depth, seed, row0=0, rows=33)
    bot, _, _ = oracle_lib.render(sc, W, H, samples, depth, seed, row0=33, rows=H - 33)
    assert np.array_equal(np.concatenate([top, bot]), z["rgb"])


def test_camera_matrices_match_glm_bits():
    with open(os.path.join(GOLDEN, "camera_bits.json")) as fh:
        cams = json.load(fh)
    for key, bits in cams.items():
        W, H = (int(v) for v in key.split("x"))
        cam = oracle_lib.camera(scenes.default4(), W, H)
        assert [int(b) for b in cam.view(np.uint32)] == bits, key
    # SURVEY.md Appendix C known answers for Camera(60, 1024, 512, 0.01, 1000)
    cam = oracle_lib.camera(scenes.default4(), 1024, 512)
    assert float(cam[3]).hex() == "0x1.279a740000000p+0"           # invProj[0][0]
    assert float(cam[3 + 5]).hex() == "0x1.279a740000000p-1"       # invProj[1][1]
    assert float(cam[3 + 16]).hex() == "0x1.ea68320000000p-1"      # invView[0][0]


def test_rng_matches_glibc_known_answers():
    with open(os.path.join(GOLDEN, "glibc_rand.json")) as fh:
        known = json.load(fh)
    for seed, seq in known.items():
        assert oracle_lib.rand_sequence(int(seed), len(seq)) == seq, seed
    assert known["1"][0] == 1804289383 and known["0"] == known["1"]      # srand(0) == srand(1)


def test_rng_matches_live_libc_on_random_seeds():
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand.argtypes = [ctypes.c_uint]
    rs = np.random.RandomState(0)
    for seed in [int(s) for s in rs.randint(0, 2**32, size=40, dtype=np.uint64)]:
        libc.srand(seed)
        want = [libc.rand() for _ in range(400)]
        assert oracle_lib.rand_sequence(seed, 400) == want, seed


def test_quantise_is_double_sqrt_truncation():
    # structs.h:12-17: (unsigned char)(sqrt((double)c) * 255.0)
    rs = np.random.RandomState(1)
    for c in list(rs.uniform(0, 1, 2000).astype(np.float32)) + [0.0, 1.0, 0.25, np.float32(1.0) - np.float32(2**-24)]:
        assert oracle_lib.quantise(c) == int(np.sqrt(np.float64(np.float32(c))) * 255.0)
    assert oracle_lib.quantise(float("nan")) == 0


def test_probe_mode_matches_reference():
    z = np.load(os.path.join(GOLDEN, "probe_glass_mirror_128x64_d3_seed0.npz"))
    sc = scenes.Scene.from_text(bytes(z["scene"]).decode())
    W, H, _, depth, seed = (int(v) for v in z["params"])
    assert np.array_equal(oracle_lib.probe_max_depth(sc, W, H, depth, seed), z["flags"])


def test_scene_text_round_trip():
    for name, builder in scenes.BUILTIN.items():
        sc = builder()
        again = scenes.Scene.from_text(sc.to_text())
        # repr(): the NaN radius of the nan scenes compares unequal to itself
        assert repr((again.materials, again.hittables, again.lights)) == repr((sc.materials, sc.hittables, sc.lights)), name
    with pytest.raises(ValueError):
        scenes.Scene.from_text("sphere 1 2 3\n")
    with pytest.raises(ValueError):
        scenes.Scene.from_text("torus 1 2 3\n")


def test_scene_text_reads_rows_of_the_reference_tables():
    """f2: rows of the reference's material / sphere / directional_light tables (src/postgres.h:183-228) with VEC3 columns
    in readVec3's "(x, y, z)" syntax (src/postgres.h:232-256) load as the scene the reference builds from them."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = scenes.Scene.load(os.path.join(root, "scenes", "default4_tables.rtscene"))
    b = scenes.default4()
    assert (a.materials, a.hittables, a.lights) == (b.materials, b.hittables, b.lights)
    # the reference's own INSERT syntax and a psql-style dump; rows of other scenes are skipped
    c = scenes.Scene.from_text("material 0, '(0.8, 0.8, 0.0)'::VEC3, 0.0, 0.0, 0.0, '(0.0, 0.0, 0.0)'::VEC3, 0.0\n"
                               "sphere 1 | (0,1,-4) | 1 | 0 | 0\nsphere 2 | (5,1,-4) | 1 | 0 | 3\n"
                               "directional_light 0 (-0.25, -0.75, 0.45) 0\ndirectional_light 1 (1, 0, 0) 3\n")
    assert c.materials == b.materials[:1] and len(c.hittables) == 1 and c.hittables[0][1] == (0.0, 1.0, -4.0) and c.lights == b.lights
    assert len(scenes.Scene.from_text("sphere 2 (5,1,-4) 1 0 3\n", scene_id=3).hittables) == 1
    for bad in ("sphere 1 (0, 1) 1 0 0\n", "material 0 (1, 1, 1) 0 0\n", "directional_light (0, 0, 1)\n"):
        with pytest.raises(ValueError):
            scenes.Scene.from_text(bad)

