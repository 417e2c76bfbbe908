"""The C++ host side: API-compatible headers (raytracer_amd/host) + the demo driver built on them."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "raytracer_amd", "host")
EXE = os.path.join(HOST, "rt_render")

REFERENCE_STYLE_TU = r'''
// written the way the reference's tests/speedUp.cpp is: compile-time frame macros, then Render(...)
#define WIDTH 256
#define HEIGHT 128
#define SAMPLES 20
#define MAXDEPTH 20
#include "renderer.h"
#include "lights/directionalLight.h"
#include "lights/lightsList.h"
#include "hittables/hittablesList.h"
#include "hittables/sphere.h"
#include "hittables/cube.h"
#include "hittables/plane.h"
int main()
{
    Camera camera(60.0f, WIDTH, HEIGHT, 0.01f, 1000.0f);
    int NUM_LIGHTS = 1;
    Light** l_light = new Light*[NUM_LIGHTS];
    l_light[0] = new DirectionalLight({ -0.35f, -0.55f, -0.5f });
    Light* lights = new LightsList(l_light, NUM_LIGHTS);
    Material* materials = new Material[4];
    materials[0] = Material{ { 0.8f, 0.8f, 0.8f }, 0.9f, 0.75f, 0.0f, { 0.0f, 0.0f, 0.0f }, 0.0f };
    int NUM_SPHERES = 3;
    Hittable** l_world = new Hittable*[NUM_SPHERES];
    l_world[0] = new Sphere({ 0.0f, -10000.0f, 0.0f }, 10000.0f, 0);
    l_world[1] = new Plane(glm::vec3{ 0.0f, 0.0f, -9.0f }, glm::vec3{ 0.0f, 0.0f, 1.0f }, 0);
    l_world[2] = new Cube({ 4.0f, 1.75f, -1.0f }, { 1.0f, 1.0f, 1.0f }, 0);
    Hittable* world = new HittablesList(l_world, NUM_SPHERES);
    double ms = Render(camera, world, lights, materials, 4);
    glm::vec3 n = glm::normalize(glm::vec3{ 1.0f, 2.0f, 2.0f });
    delete lights; delete world; delete[] materials; delete[] l_light; delete[] l_world;
    return ms < 0.0 || n.x > 1.0f;
}
'''


def test_reference_style_driver_compiles_against_host_headers():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "speedup_like.cpp")
        with open(src, "w") as fh:
            fh.write(REFERENCE_STYLE_TU)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", HOST, "-I", os.path.join(ROOT, "include"), src],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def _read_ppm(path):
    with open(path, "rb") as fh:
        data = fh.read()
    magic, dims, maxv, body = data.split(b"\n", 3)
    w, h = (int(v) for v in dims.split())
    assert magic == b"P6" and maxv == b"255"
    return np.frombuffer(body, dtype=np.uint8).reshape(h, w, 3)


@pytest.mark.gpu
def test_cpp_driver_default_scene_matches_golden(golden_dir):
    assert os.path.exists(EXE), "build raytracer_amd/host/rt_render (make -C raytracer_amd/host)"
    z = np.load(os.path.join(golden_dir, "default4_256x128_s5_d10_seed1000.npz"))
    with tempfile.TemporaryDirectory() as td:
        out, em = os.path.join(td, "o.ppm"), os.path.join(td, "o.em")
        r = subprocess.run([EXE, "--w", "256", "--h", "128", "--samples", "5", "--depth", "10", "--seed-base", "1000",
                            "--out", out, "--emission", em], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(_read_ppm(out), z["rgb"])
        assert np.array_equal(np.fromfile(em, dtype=np.uint32), z["emission"].view(np.uint32).ravel())


@pytest.mark.gpu
def test_cpp_driver_ranks_mode_matches_golden(golden_dir):
    """--ranks 1: RenderImageRanks() -> rt_render_gather (band render, rt_gather_rows with a one-rank RCCL communicator,
    reassembly) gives the golden frame.  More ranks fork one process per GPU: not runnable on a one-GPU box."""
    z = np.load(os.path.join(golden_dir, "default4_256x128_s5_d10_seed1000.npz"))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.ppm")
        r = subprocess.run([EXE, "--w", "256", "--h", "128", "--samples", "5", "--depth", "10", "--seed-base", "1000",
                            "--out", out, "--ranks", "1", "--frames", "2"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "1 ranks" in r.stdout
        assert np.array_equal(_read_ppm(out), z["rgb"])


@pytest.mark.gpu
def test_cpp_driver_reads_reference_table_rows(golden_dir):
    """f2: LoadScene() reads rows of the reference's tables with "(x, y, z)" VEC3 columns (src/postgres.h:183-256)."""
    z = np.load(os.path.join(golden_dir, "default4_256x128_s5_d10_seed1000.npz"))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.ppm")
        r = subprocess.run([EXE, "--scene", os.path.join(ROOT, "scenes", "default4_tables.rtscene"), "--w", "256", "--h", "128",
                            "--samples", "5", "--depth", "10", "--seed-base", "1000", "--out", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(_read_ppm(out), z["rgb"])


@pytest.mark.gpu
def test_cpp_driver_glow_matches_the_cpu_restatement(golden_dir):
    """--glow: RenderImage() then ApplyGlow() (src/main.cpp:400-406 order), against oracle/rt_glow.c on the golden frame."""
    import oracle_lib
    z = np.load(os.path.join(golden_dir, "default4_256x128_s5_d10_seed1000.npz"))
    want, levels = oracle_lib.apply_glow(z["rgb"], z["emission"])
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.ppm")
        r = subprocess.run([EXE, "--w", "256", "--h", "128", "--samples", "5", "--depth", "10", "--seed-base", "1000",
                            "--out", out, "--glow"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Glow: %d levels" % levels in r.stdout
        assert np.array_equal(_read_ppm(out), want)


@pytest.mark.gpu
def test_cpp_driver_scene_file_and_probe(golden_dir):
    z = np.load(os.path.join(golden_dir, "cube_128x64_s4_d10_seed0.npz"))
    with tempfile.TemporaryDirectory() as td:
        sp, out = os.path.join(td, "s.rtscene"), os.path.join(td, "o.ppm")
        with open(sp, "wb") as fh:
            fh.write(bytes(z["scene"]))
        r = subprocess.run([EXE, "--scene", sp, "--w", "128", "--h", "64", "--samples", "4", "--depth", "10", "--out", out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(_read_ppm(out), z["rgb"])
        zp = np.load(os.path.join(golden_dir, "probe_glass_mirror_128x64_d3_seed0.npz"))
        with open(sp, "wb") as fh:
            fh.write(bytes(zp["scene"]))
        r = subprocess.run([EXE, "--scene", sp, "--w", "128", "--h", "64", "--depth", "3", "--probe"], capture_output=True, text=True)
        assert r.returncode == 0 and "Status: Failed!" in r.stdout      # some pixel reaches the cap in this scene
        r = subprocess.run([EXE, "--w", "128", "--h", "64", "--samples", "2", "--speedup"], capture_output=True, text=True)
        assert r.returncode == 0 and "Ended in:" in r.stdout


@pytest.mark.gpu
def test_cpp_driver_tile_stream_recomposes_golden(golden_dir):
    """--tiles: RenderTiles() hands tiles to a SendImage-shaped callback (src/redis.h:27), the driver recomposes them
    like the reference's consumer loop (src/main.cpp:378-395) and prints the reference's progress line."""
    z = np.load(os.path.join(golden_dir, "default4_256x128_s5_d10_seed1000.npz"))
    with tempfile.TemporaryDirectory() as td:
        out, em = os.path.join(td, "o.ppm"), os.path.join(td, "o.em")
        r = subprocess.run([EXE, "--w", "256", "--h", "128", "--samples", "5", "--depth", "10", "--seed-base", "1000",
                            "--tiles", "16", "--chunk-rows", "32", "--out", out, "--emission", em], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(_read_ppm(out), z["rgb"])
        assert np.array_equal(np.fromfile(em, dtype=np.uint32), z["emission"].view(np.uint32).ravel())
        lines = [l for l in r.stdout.splitlines() if l.startswith("Progress:")]
        assert len(lines) == 4 and lines[-1].startswith("Progress: 100.0")


@pytest.mark.gpu
def test_cpp_driver_frames_in_flight(golden_dir):
    """--frames: RenderFrames() keeps several frames under way (one context per producer thread) and delivers them in
    order; frame 1 of a sequence seeded 0, 1000, 2000, ... is the golden seed-1000 frame, and every frame equals the
    same frame rendered alone."""
    z = np.load(os.path.join(golden_dir, "default4_256x128_s5_d10_seed1000.npz"))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "seq")
        r = subprocess.run([EXE, "--w", "256", "--h", "128", "--samples", "5", "--depth", "10", "--seed-base", "0",
                            "--frames", "7", "--in-flight", "3", "--out", out], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "out of order" not in r.stderr, r.stdout + r.stderr
        assert np.array_equal(_read_ppm(out + ".1.ppm"), z["rgb"])
        for k in (0, 4, 6):
            alone = os.path.join(td, "alone.ppm")
            r = subprocess.run([EXE, "--w", "256", "--h", "128", "--samples", "5", "--depth", "10", "--seed-base", str(1000 * k),
                                "--out", alone], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            assert np.array_equal(_read_ppm(out + ".%d.ppm" % k), _read_ppm(alone)), k
