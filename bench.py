#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary closest-hit queries) of the TraceRay hot path.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step renders one whole frame of the workload: every rank renders its cyclic row bands with rt_render_rows into a
device buffer, the bands are gathered to rank 0 (RCCL) and reassembled there.  N = 1 workload: BASELINE.json's metric
configuration, DEFAULT-4 at 3840x2160, 64 spp, max depth 10 (it fits one GPU).  The frame is fixed, so scaling is
"strong".  Ray counts are exact: the counting build of the kernel tallies closest-hit queries for the same frame
outside the timed region (they equal the reference's, tests/test_gpu_parity.py).

Frames in flight.  A rank's share of a frame is bound by the latency of its longest pixels, not by throughput
(3.3 ms for an eighth of the 4K frame whose arithmetic is worth 0.7 ms), so the K timed frames are
rendered with up to --frames-in-flight (default 8, at every N) frames of a rank under way at once
(raytracer_amd/pipeline.py: host threads with a context each; gathers in frame order).  Every frame is rendered,
gathered and reassembled in full inside the timed region.  After it, three frames run one at a time:
`frame_latency_ms` is the time of such a frame, and the roofline object prices the render kernels of such a frame
(HIP events around them; with frames in flight the kernels of different frames overlap and a launch's duration says
nothing), which is also what the rocprofv3 summaries under profiles/ hold (`--frames-in-flight 1`).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# frames in flight need a hardware queue per stream that is busy at once (3 per context); the box's default of 4 makes
# streams of different frames wait for each other.  Must be set before HIP starts.  (--frames-in-flight 1 leaves it.)
def _frames_in_flight_arg(argv):
    gpus, fif = 1, None
    for i, a in enumerate(argv):
        for name in ("--gpus", "--frames-in-flight"):
            v = None
            if a == name and i + 1 < len(argv):
                v = argv[i + 1]
            elif a.startswith(name + "="):
                v = a.split("=", 1)[1]
            if v is not None and v.lstrip("-").isdigit():
                if name == "--gpus":
                    gpus = int(v)
                else:
                    fif = int(v)
    return fif if fif else DEFAULT_IN_FLIGHT


DEFAULT_IN_FLIGHT = 16
HW_QUEUES = os.environ.get("RT_HW_QUEUES", "16")
if _frames_in_flight_arg(sys.argv[1:]) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = HW_QUEUES

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene builder name, W, H, samples, max_depth)
    "default4_4k_64spp": ("default4", 3840, 2160, 64, 10),       # BASELINE.json configs[2] / metric
    "default4_1080p_16spp": ("default4", 1920, 1080, 16, 5),     # configs[1]
    "default4_1024x512_5spp": ("default4", 1024, 512, 5, 10),    # configs[0]
    "many2048_1080p_32spp": ("many_objects", 1920, 1080, 32, 10),  # configs[3]
    "glass_1080p_64spp": ("glass_mirror", 1920, 1080, 64, 12),   # configs[4]
    # f1: the glow post-process of the reference's consumer (src/main.cpp:163-229) on a rendered 4K frame; a step = one
    # rt_apply_glow over frame + emission buffer resident in HBM (run_glow below; N = 1 only: it is rank 0's step)
    "glow_4k": ("default4", 3840, 2160, 4, 10),
}
PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
BAND_ROWS = 16


def algorithmic_flops(cnt, scene, n_pixels, samples):
    """SURVEY.md section 8(d) per-op table applied to the frame's query tallies as the reference algorithm issues them
    (the device counting build, equal to the oracle's tallies) -- the algorithmic work of the frame, not what the
    production kernels execute: they skip tests that are known to fail.
    ray generation 71 + per-ray accumulate 10; primitive test 30 sphere / 14 plane / 28 cube (type-weighted by the
    scene's mix, exact for single-type scenes); per shaded hit: epilogue 19 + shade 17 + 6 per light; sky 15;
    bounce 76 (the cheaper of reflection 76 / refraction 95, the tallies do not split them); per-sample clamp/add 19;
    per-pixel finish 9."""
    types = [h[0] for h in scene.hittables]
    per_test = (30.0 * types.count(0) + 14.0 * types.count(1) + 28.0 * types.count(2)) / max(len(types), 1)
    n_lights = max(len(scene.lights), 1)
    hits = cnt["shaded"]
    sky = cnt["closest"] - hits
    bounces = cnt["rand_draws"] / 3.0
    return (81.0 * cnt["primary"] + per_test * cnt["tests"] + hits * (19.0 + 17.0 + 6.0 * n_lights) + 15.0 * sky +
            76.0 * bounces + 19.0 * n_pixels * samples + 9.0 * n_pixels)


def host_threads():
    """Worker threads for the CPU legs: hardware_concurrency() - 1 like src/main.cpp:268, where the hardware is what this
    process may use -- its CPU affinity set, and on a GPU box the share a one-GPU job is given (os.cpu_count() reports the
    whole machine there; the box exports its share as OMP_NUM_THREADS and kills jobs that start more workers)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n - 1)


def kernel_source_hash():
    """sha256 over the sources and build flags the kernels are made of (raytracer_amd/csrc/*.hip|hpp|h|cpp + Makefile, in
    name order).  tools/pmc_frame.py stores it in profiles/pmc_<workload>.json; a profile whose hash is not this tree's was
    counted on other kernels, and the roofline object then says so instead of pricing this run with it."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.h")) +
                    glob.glob(os.path.join(d, "*.cpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load_pmc(workload):
    """profiles/pmc_<workload>.json (tools/pmc_frame.py over the rocprofv3 --pmc passes of this bench command): executed VALU
    wave instructions, executed FP32 operations and HBM bytes of one frame.  -> (profile, path, stale): no profile: (None,
    None, False); stale = the profile was taken on other kernel sources than this tree's (kernel_source_hash)."""
    path = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)
    if not os.path.exists(path):
        return None, None, False
    with open(path) as fh:
        prof = json.load(fh)
    return prof, os.path.relpath(path, ROOT), prof.get("kernel_source_sha256") != kernel_source_hash()


def cpu_baseline(scene, W, H, samples, depth, count_rays, budget_s=20.0):
    """Times the reference's own multi-threaded renderer (oracle/_ref/ref_oracle: the unmodified reference headers
    driven through the reference's ThreadPool with 16x16 tiles, src/main.cpp:268-331) on this box's host cores, on a
    bounded sample of the same workload: every row_step-th row of the frame at the full spp.  count_rays(row_step)
    returns the exact closest-hit query count of those rows.  Falls back to the plain-C port (kind "port") when the
    reference binary is absent; when both exist the port's rate is reported alongside."""
    cores = host_threads()
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_oracle")
    port = os.path.join(ROOT, "oracle", "rt_oracle")
    if not os.path.exists(port):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, capture_output=True)
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        sp = os.path.join(td, "scene.rtscene")
        scene.save(sp)
        for kind, exe in (("reference", ref), ("port", port)):
            if not os.path.exists(exe):
                continue
            step = 64
            while True:
                r = subprocess.run([exe, "--scene", sp, "--w", str(W), "--h", str(H), "--samples", str(samples), "--depth", str(depth),
                                    "--seed-base", "0", "--threads", str(cores), "--row-step", str(step)], capture_output=True, text=True, check=True)
                info = json.loads(r.stdout.strip().splitlines()[-1])
                best = {"ms": info["ms"], "step": step, "rows": info["rows"]}
                if info["ms"] * 4 > budget_s * 1e3 / 2 or step <= 4:
                    break
                step //= 4
            best["value"] = count_rays(best["step"]) / (best["ms"] * 1e-3) / 1e6
            out[kind] = best
        # the reference's own build line has no -O flag (`g++ -g`, run_cpu.sh:1): the same harness built that way, on a
        # quarter of the rows
        ref_o0 = os.path.join(ROOT, "oracle", "_ref", "ref_oracle_O0")
        if "reference" in out and os.path.exists(ref_o0):
            step = out["reference"]["step"] * 4
            r = subprocess.run([ref_o0, "--scene", sp, "--w", str(W), "--h", str(H), "--samples", str(samples), "--depth", str(depth),
                                "--seed-base", "0", "--threads", str(cores), "--row-step", str(step)], capture_output=True, text=True, check=True)
            info = json.loads(r.stdout.strip().splitlines()[-1])
            out["reference_O0"] = {"ms": info["ms"], "step": step, "rows": info["rows"], "value": count_rays(step) / (info["ms"] * 1e-3) / 1e6}
    kind = "reference" if "reference" in out else "port"
    b = out[kind]
    res = {"value": round(b["value"], 3), "unit": "Mrays/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": kind,
           "flags": "g++ -O2 -std=c++17 -ffp-contract=off -pthread (oracle/Makefile: the reference's headers, unmodified)" if kind == "reference"
                    else "gcc -O2 -std=c11 -ffp-contract=off (oracle/Makefile)",
           "sample": "every %d-th row (%d rows) of the %dx%d frame at %d spp, %.0f ms, 16x16 tiles on %d worker threads" % (
               b["step"], b["rows"], W, H, samples, b["ms"], cores)}
    if "reference_O0" in out:
        o0 = out["reference_O0"]
        res["reference_own_flags"] = {"flags": "g++ -g -std=c++17 -pthread (the reference's own: run_cpu.sh:1, no -O)", "value": round(o0["value"], 3),
                                      "sample": "every %d-th row (%d rows), %.0f ms" % (o0["step"], o0["rows"], o0["ms"])}
    if kind == "reference" and "port" in out:
        res["port_value"] = round(out["port"]["value"], 3)
        res["port_sample_ms"] = round(out["port"]["ms"], 1)
    return res


def glow_algorithmic_bytes(W, H, levels):
    """HBM bytes one applyGlow has to move: per level l (factor r = 2^l, n = W H / r^2 level pixels of 16 B) the
    downsample reads its source (the frame's emission at l = 1, the previous level after) and writes n; the blur reads and
    writes n (its taps come from cache); the add reads n and reads + writes the frame's 3 B pixels; the attenuation reads
    and writes n."""
    npix, total, src = W * H, 0, W * H
    for l in range(1, levels + 1):
        n = npix // (4 ** l)
        total += 16 * src + 16 * n + 32 * n + 16 * n + 6 * npix + 32 * n
        src = n
    return total


def run_glow(args):
    """--workload glow_4k: rt_apply_glow (f1) timed on device buffers, checked against oracle/rt_glow.c (a restatement:
    parity unpinned) on a 1024x512 frame, with the restatement's own time beside it."""
    import numpy as np
    import torch
    import raytracer_amd as ra
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1:
        raise SystemExit("bench.py: glow_4k is the consumer's step on rank 0 (src/main.cpp:402): run it with --gpus 1")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    scene_name, W, H, samples, depth = WORKLOADS[args.workload]
    scene = ra.scenes.BUILTIN[scene_name]()
    rt = ra.Renderer(0)
    rt.upload_scene(scene)

    def rendered(w, h):
        rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        em = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        rt.render_rows(ra.camera35f(scene, w, h), rt.params(w, h, samples, depth, 0), rgb, em)
        return rgb, em

    # parity first, at a size the CPU restatement finishes in seconds
    rgb_s, em_s = rendered(1024, 512)
    t0 = time.perf_counter()
    want, want_levels = oracle_lib.apply_glow(rgb_s.cpu().numpy(), em_s.cpu().numpy())
    cpu_s = time.perf_counter() - t0
    got_levels = rt.apply_glow(rgb_s, em_s, 1024, 512)
    bad = int((rgb_s.cpu().numpy() != want).any(axis=-1).sum())
    verified = {"frame": "1024x512", "levels": int(got_levels), "oracle_levels": int(want_levels), "differing_pixels": bad,
                "identical": bad == 0 and got_levels == want_levels, "against": "oracle/rt_glow.c (restatement, parity unpinned)"}

    base, em = rendered(W, H)
    rgb = base.clone()
    ms = []
    levels = 0
    for i in range(args.warmup + args.steps):
        rgb.copy_(base)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        levels = rt.apply_glow(rgb, em, W, H)               # (returns when the frame is done)
        if i >= args.warmup:
            ms.append((time.perf_counter() - t0) * 1e3)
    t = float(np.mean(ms))
    alg = glow_algorithmic_bytes(W, H, levels)
    pmc, pmc_path, pmc_stale = load_pmc(args.workload)
    line = {
        "metric": "Mpixels/s through applyGlow (src/main.cpp:163-229) at 4K", "value": round(W * H / (t * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H, "spp": samples, "levels": int(levels)},
        "verified": verified,
        "roofline": {"bound": "hbm", "achieved": round(alg / (t * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(alg / (t * 1e-3) / 1e9 / 8000.0, 4), "traffic": int(pmc["hbm_bytes"]) if pmc and not pmc_stale else None,
                     "algorithmic_bytes": alg, "pmc": pmc_path, "pmc_stale": bool(pmc_stale),
                     "note": "whole rt_apply_glow call (4 kernels and one 4-byte read-back per level), wall time; algorithmic bytes per "
                             "glow_algorithmic_bytes(); the call is bound by its per-level launch / read-back chain and the blur's "
                             "arithmetic (2.2e9 taps per 4K frame), not by HBM"},
        "cpu_baseline": {"value": round(1024 * 512 / cpu_s / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                         "sample": "oracle/rt_glow.c on the 1024x512 frame, %.1f s" % cpu_s},
    }
    print(json.dumps(line), flush=True)
    rt.close()
    if not verified["identical"]:
        raise SystemExit("bench.py: rt_apply_glow differs from oracle/rt_glow.c")


def main():
    t_start = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-in-flight", type=int, default=0, help="frames of one rank under way at once (default %d)" % DEFAULT_IN_FLIGHT)
    ap.add_argument("--workload", default="default4_4k_64spp", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--estimate-tally", action="store_true", help="tally every 32nd row and scale instead of the whole frame (large scenes: the exact tally takes most of a minute)")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the timed frame with the CPU oracle")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs after the headline: the relaxed-arithmetic variant of the same workload, the glass 1080p frame and the 2049-hittable 1080p frame")
    ap.add_argument("--flags", type=int, default=0, help="extra rt_params.flags for A/B runs (4 = RT_FLAG_SINGLE_KERNEL, 8 = RT_FLAG_FORCE_SAMPLE_PARALLEL)")
    args = ap.parse_args()

    if args.workload.startswith("glow_"):
        return run_glow(args)
    import numpy as np
    import torch
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd import bands
    from raytracer_amd.pipeline import FramePipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    # rehearsal knobs for a one-GPU box (RCCL refuses two ranks on one device): RT_BENCH_DEVICE pins every rank to one
    # device, RT_BENCH_BACKEND=gloo swaps the transport; the control flow -- pipeline, gather order, reductions -- is
    # the one the N-GPU job runs
    if os.environ.get("RT_BENCH_DEVICE"):
        local_rank = int(os.environ["RT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The band gather goes through the library's rt_gather_rows (RCCL on the consumer context's stream); torch.distributed
    # only carries the communicator's unique id and the timing reductions, for which gloo does.  RT_BENCH_GATHER=torch:
    # torch.distributed.gather instead (backend nccl = RCCL too; gloo for the one-GPU rehearsal).
    gather_mode = os.environ.get("RT_BENCH_GATHER", "rt")
    backend = os.environ.get("RT_BENCH_BACKEND", "gloo" if gather_mode == "rt" else "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    rdev = dev if backend == "nccl" else torch.device("cpu")       # where the reductions' tensors live

    scene_name, W, H, samples, depth = WORKLOADS[args.workload]
    scene = ra.scenes.BUILTIN[scene_name]()
    seed = 0
    in_flight = max(1, min(_frames_in_flight_arg(sys.argv[1:]), max(args.steps, 1)))
    if len(scene.hittables) >= 256 and "--frames-in-flight" not in " ".join(sys.argv[1:]):
        # a large scene's frame fills the GPU by itself (coop_kernel: one wave per bounce pixel from a queue), and its
        # workgroups want an empty GPU to spread evenly over the SIMDs: frames one at a time
        # (many2048 1080p/32spp: 0.84 s per frame against 0.92 s with three in flight)
        in_flight = 1
    # one host thread per frame in flight waits inside rt_render_rows most of the time: under RT_FLAG_SEQUENCE the
    # library polls an event with short naps there ("nap") instead of HIP's busy-polling wait ("spin")
    host_waits = "nap" if in_flight > 1 else "spin"
    rts = [ra.Renderer(local_rank) for _ in range(in_flight)]
    for r in rts:
        r.upload_scene(scene)
    rt = rts[0]
    consumer = ra.Renderer(local_rank)       # gathers and reassembles: a context (stream) of its own, beside the producers'
    # frames rendered one at a time (warm-up, latency, the kernel span the roofline object prices) are not a sequence:
    # they run without RT_FLAG_SEQUENCE, on a context of their own (created last: its streams are idle in the timed region)
    rt_solo = rt
    if in_flight > 1:
        rt_solo = ra.Renderer(local_rank)
        rt_solo.upload_scene(scene)
    comm, gather_group = None, None
    if world > 1 and gather_mode == "rt":
        idt = torch.zeros(ra._abi.RT_COMM_ID_BYTES, dtype=torch.uint8, device=rdev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(ra.Renderer.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        ok, why = 1, ""
        try:
            comm = (consumer.comm_create(bytes(idt.cpu().numpy().tobytes()), rank, world), rank, world)
        except Exception as e:          # (reported in the JSON line; every rank then takes the torch.distributed gather)
            ok, why = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=rdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm:
                consumer.comm_destroy(comm[0])
            comm = None
            gather_mode = "torch (rt_gather_rows unavailable: %s)" % (why or "another rank failed")
            gather_group = dist.new_group(backend="nccl", device_id=dev) if backend != "nccl" else None
    cam = ra.camera35f(scene, W, H)
    rpr = bands.rows_per_rank(H, BAND_ROWS, world)
    seq = ra._abi.RT_FLAG_SEQUENCE if in_flight > 1 else 0          # several frames in flight on this GPU
    p = bands.band_params(rt, W, H, samples, depth, seed, BAND_ROWS, world, rank, flags=args.flags | seq)
    p_solo = bands.band_params(rt_solo, W, H, samples, depth, seed, BAND_ROWS, world, rank, flags=args.flags)
    blocks = [torch.zeros((rpr, W, 3), dtype=torch.uint8, device=dev) for _ in range(in_flight * (2 if in_flight > 1 else 1))]
    block = blocks[0]
    torch.cuda.synchronize()        # the library renders on its own stream: the zero fill must have landed
    kernel_ms, device_ms = [], []

    def render(worker, out, k):
        rts[worker].render_rows(cam, p, out)
        if worker == 0:
            kernel_ms.append(rt.last_kernel_ms)
            device_ms.append(rt.last_device_ms)

    def sequence(cam_, p_, blocks_, H_, frames):
        """`frames` frames of another parameter set through the same contexts, in flight like the headline: -> (seconds
        per frame, the last frame).  One untimed round first."""
        def render_(worker, out, k):
            rts[worker].render_rows(cam_, p_, out)
        pp = FramePipeline(render_, blocks_, H_, BAND_ROWS, consumer, group=gather_group, workers=in_flight, comm=comm)
        pp.run(min(in_flight, frames))
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        f_ = pp.run(frames)
        torch.cuda.synchronize()
        return (time.perf_counter() - t_) / frames, f_

    def render_solo(worker, out, k):
        rt_solo.render_rows(cam, p_solo, out)
        kernel_ms.append(rt_solo.last_kernel_ms)
        device_ms.append(rt_solo.last_device_ms)

    pipe = FramePipeline(render, blocks, H, BAND_ROWS, consumer, group=gather_group, workers=in_flight, comm=comm)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: W frames one at a time, then every context once
    solo = FramePipeline(render_solo if in_flight > 1 else render, blocks[:1], H, BAND_ROWS, consumer, group=gather_group, workers=1, comm=comm)
    for _ in range(args.warmup):
        solo.run(1)
    if in_flight > 1:
        # (untimed) every context several times, then the timed region's own pattern once: the K timed frames take some
        # 80 ms, and a one-off inside them -- a hardware queue's scratch ring growing at a context's first listed kernel,
        # clocks still ramping on a fresh box -- showed as 6.1 instead of 3.9 ms per frame in one run of two
        pipe.run(4 * in_flight)
        pipe.run(args.steps)
    sync()
    del kernel_ms[:], device_ms[:]
    import gc
    gc.collect()
    gc.disable()                                    # (no collector pause among the host threads of the timed region)
    t0 = time.perf_counter()
    frame = pipe.run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    gather_ms = list(pipe.gather_ms)                 # per frame: this rank's gather + reassembly step
    # untimed: the last frame of the timed region against the CPU oracle (the plain-C port, pinned to the reference by
    # tests/test_oracle_golden.py) on every vstep-th row -- the rows the cpu_baseline leg samples
    def progress(what):              # (stderr: the one JSON line stays alone on stdout; a silent run reads as hung)
        if rank == 0:
            print("[bench] %s (%.0f s)" % (what, time.perf_counter() - t_start), file=sys.stderr, flush=True)

    progress("timed region done: %d frame(s) in %.3f s" % (args.steps, elapsed))
    verified, want, vstep = None, None, 16
    if rank == 0 and not args.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        vstep = 16 if len(scene.hittables) < 256 else H // 3       # (large scenes: minutes per row on the CPU)
        vrows = (H + vstep - 1) // vstep
        tv0 = time.perf_counter()
        want = oracle_lib.render_strided(scene, W, H, samples, depth, seed, row0=0, rows=vrows, row_step=vstep, threads=host_threads())[0]
        got = frame[::vstep].cpu().numpy()
        bad = int((got != want).any(axis=-1).sum())
        verified = {"rows": vrows, "row_step": vstep, "pixels": int(vrows * W), "differing_pixels": bad, "identical": bad == 0,
                    "against": "oracle/librt_oracle.so (CPU port), %.1f s" % (time.perf_counter() - tv0)}
    # untimed: three frames one at a time -- the latency of a frame on its own and the durations of its kernels when
    # they do not overlap with another frame's (what the roofline object prices)
    latency_ms = []
    if in_flight > 1:
        del kernel_ms[:], device_ms[:]
    for _ in range(3):
        sync()
        t1 = time.perf_counter()
        solo.run(1)
        sync()
        latency_ms.append((time.perf_counter() - t1) * 1e3)
    if in_flight == 1:                       # the timed frames themselves ran one at a time
        del kernel_ms[args.steps:], device_ms[args.steps:]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    progress("verification done" if verified is not None else "verification skipped")
    # exact tallies of this frame (untimed): the counting build issues every query the reference issues
    pc = bands.band_params(rt, W, H, samples, depth, seed, BAND_ROWS, world, rank)
    tally_note = "queries of the reference algorithm for this frame, tallied by the device's counting build (equal to the oracle's tallies: tests/test_gpu_parity.py, tests/test_gpu_fullsize.py)"
    if args.estimate_tally:
        # the counting build is one lane per pixel over the whole scene: most of a minute for a 1080p frame of a
        # 2049-hittable scene.  On request, tally every 32nd row of this rank's rows and scale -- an estimate, flagged
        # as such.
        step = 32
        n = max(1, p.rows // step)
        ps = rt.params(W, H, samples, depth, seed, row0=pc.row0, rows=n, band_rows=1, band_stride=step * world)
        cnt = rt.render_rows(cam, ps, block[:n], count=True)
        cnt = {k: int(v * (p.rows / float(n))) for k, v in cnt.items()}
        tally_note = "estimate: reference-algorithm queries of every %d-th row, scaled to the frame" % step
    else:
        cnt = rt.render_rows(cam, pc, block, count=True)
    progress("tally done")
    keys = ("primary", "closest", "shadow", "tests", "rand_draws", "shaded")
    tv = torch.tensor([cnt[k] for k in keys], dtype=torch.int64, device=rdev)
    local_flops = algorithmic_flops(cnt, scene, W * p.rows, samples)
    # duration of one frame's render kernels when they do not overlap with another frame's
    lf = torch.tensor([local_flops, float(np.mean(kernel_ms))], dtype=torch.float64, device=rdev)
    kms_min = float(lf[1].item())
    if world > 1:
        dist.all_reduce(tv)
        mx = lf.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        mn = lf.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(lf)
        flops, kms, kms_min = float(lf[0].item()), float(mx[1].item()), float(mn[1].item())
    else:
        flops, kms = float(lf[0].item()), float(lf[1].item())
    tot = dict(zip(keys, (int(v) for v in tv.tolist())))

    def relaxed_leg():
        """The same workload with RT_FLAG_RELAXED (include/rt_amd.h): not bit-identical, specified to the north star's
        tolerance.  Same frames in flight, same ray count (the reference algorithm's); checked against the oracle rows of
        the verification above with that tolerance."""
        from raytracer_amd._abi import RT_FLAG_RELAXED
        p_r = bands.band_params(rt, W, H, samples, depth, seed, BAND_ROWS, world, rank, flags=args.flags | seq | RT_FLAG_RELAXED)
        sec, f_r = sequence(cam, p_r, blocks, H, args.steps)
        leg = {"flag": "RT_FLAG_RELAXED", "ms_per_step": round(sec * 1e3, 3), "value": round(tot["closest"] / sec / 1e6, 2), "unit": "Mrays/s",
               "steps": args.steps, "frames_in_flight": in_flight, "bit_identical": False,
               "tolerance": ">= 99.9 % of the 8-bit pixels within 1 LSB of the reference's, >= 50 dB PSNR (BASELINE.json north star)"}
        if want is not None:
            d = np.abs(f_r[::vstep].cpu().numpy().astype(np.int16) - want.astype(np.int16))
            per = d.max(axis=-1)
            mse = float((d.astype(np.float64) ** 2).mean())
            beyond = int((per > 1).sum())
            leg["verified"] = {"pixels": int(per.size), "differing_pixels": int((per > 0).sum()), "pixels_beyond_1_lsb": beyond,
                               "within_1_lsb_frac": round(1.0 - beyond / per.size, 7), "max_delta": int(per.max()),
                               "psnr_db": None if mse == 0.0 else round(10.0 * float(np.log10(255.0 ** 2 / mse)), 2),
                               "within_tolerance": beyond <= 0.001 * per.size and (mse == 0.0 or 10.0 * np.log10(255.0 ** 2 / mse) >= 50.0)}
        pm, pm_path, pm_stale = load_pmc(args.workload + "_relaxed")
        if pm and not pm_stale:
            # one frame's kernels one at a time, as for the headline's roofline object
            rt_solo.render_rows(cam, bands.band_params(rt_solo, W, H, samples, depth, seed, BAND_ROWS, world, rank, flags=args.flags | RT_FLAG_RELAXED), blocks[0])
            k_ms = rt_solo.last_kernel_ms
            pf = pm["per_frame"]
            leg["roofline"] = {"kernel_ms": round(k_ms, 3), "pmc": pm_path,
                               "valu_issue_frac": round(pf["valu_wave_instructions"] / (k_ms * 1e-3) / 1e9 / (256 * 4 * 2.4 / 2.0), 4),
                               "timed_region_valu_issue_frac": round(pf["valu_wave_instructions"] / sec / 1e9 / (256 * 4 * 2.4 / 2.0), 4),
                               "executed_tflops": round(pf["executed_flops"] / (k_ms * 1e-3) / 1e12, 2),
                               "executed_frac": round(pf["executed_flops"] / (k_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
                               "fma_share": round(pf["fma_share_of_fp32_instructions"], 4)}
        elif pm:
            leg["roofline"] = {"pmc": pm_path, "pmc_stale": True}
        return leg

    def glass_leg(name="glass_1080p_64spp", frames=32):
        """BASELINE.json configs[4] (mirror / glass, 1080p, 64 spp, depth 12) through the same contexts: a driver-timed
        figure for the workload whose pixels are long.  Exact arithmetic; every 24th row against the oracle."""
        sname, W2, H2, spp2, depth2 = WORKLOADS[name]
        sc2 = ra.scenes.BUILTIN[sname]()
        for r_ in rts + [rt_solo]:
            r_.upload_scene(sc2)
        cam2 = ra.camera35f(sc2, W2, H2)
        rpr2 = bands.rows_per_rank(H2, BAND_ROWS, world)
        blocks2 = [torch.zeros((rpr2, W2, 3), dtype=torch.uint8, device=dev) for _ in range(len(blocks))]
        torch.cuda.synchronize()
        p2 = bands.band_params(rt, W2, H2, spp2, depth2, seed, BAND_ROWS, world, rank, flags=seq)
        sec, f2 = sequence(cam2, p2, blocks2, H2, frames)
        t1 = time.perf_counter()
        rt_solo.render_rows(cam2, bands.band_params(rt_solo, W2, H2, spp2, depth2, seed, BAND_ROWS, world, rank), blocks2[0])
        alone_ms = (time.perf_counter() - t1) * 1e3
        cnt2 = rt.render_rows(cam2, bands.band_params(rt, W2, H2, spp2, depth2, seed, BAND_ROWS, world, rank), blocks2[0], count=True)
        leg = {"workload": name, "config": {"scene": sname, "width": W2, "height": H2, "spp": spp2, "max_depth": depth2},
               "ms_per_step": round(sec * 1e3, 3), "value": round(cnt2["closest"] / sec / 1e6, 2), "unit": "Mrays/s", "steps": frames,
               "frames_in_flight": in_flight, "frame_latency_ms": round(alone_ms, 2),
               "rays": {"primary": cnt2["primary"], "secondary": cnt2["closest"] - cnt2["primary"], "shadow": cnt2["shadow"], "rng_draws": cnt2["rand_draws"]}}
        if not args.no_verify:
            step2 = 24
            n2 = (H2 + step2 - 1) // step2
            w2 = oracle_lib.render_strided(sc2, W2, H2, spp2, depth2, seed, row0=0, rows=n2, row_step=step2, threads=host_threads())[0]
            bad2 = int((f2[::step2].cpu().numpy() != w2).any(axis=-1).sum())
            leg["verified"] = {"rows": n2, "row_step": step2, "pixels": int(n2 * W2), "differing_pixels": bad2, "identical": bad2 == 0}
        for r_ in rts + [rt_solo]:
            r_.upload_scene(scene)
        return leg

    def large_scene_leg(name="many2048_1080p_32spp", frames=3):
        """BASELINE.json configs[3] (2049 hittables, 8 point lights, 1080p, 32 spp): a driver-timed figure for the large-scene
        route (direct_kernel<BIG> + coop_kernel).  Its frames fill the GPU by themselves: one at a time, one warm-up frame
        first.  Exact arithmetic; three rows against the oracle (minutes per row on the CPU beyond that); the ray count is
        the counting build's on every 32nd row, scaled (the whole frame takes it most of a minute)."""
        sname, W2, H2, spp2, depth2 = WORKLOADS[name]
        sc2 = ra.scenes.BUILTIN[sname]()
        rt_solo.upload_scene(sc2)
        cam2 = ra.camera35f(sc2, W2, H2)
        out2 = torch.zeros((H2, W2, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        p2 = rt_solo.params(W2, H2, spp2, depth2, seed)
        rt_solo.render_rows(cam2, p2, out2)
        k_ms = []
        t1 = time.perf_counter()
        for _ in range(frames):
            rt_solo.render_rows(cam2, p2, out2)
            k_ms.append(rt_solo.last_kernel_ms)
        sec = (time.perf_counter() - t1) / frames
        step = 32
        n = H2 // step
        cnt2 = rt_solo.render_rows(cam2, rt_solo.params(W2, H2, spp2, depth2, seed, row0=0, rows=n, band_rows=1, band_stride=step), out2.clone()[:n], count=True)
        closest = int(cnt2["closest"] * (H2 / float(n)))
        leg = {"workload": name, "config": {"scene": sname, "width": W2, "height": H2, "spp": spp2, "max_depth": depth2, "hittables": len(sc2.hittables), "lights": len(sc2.lights)},
               "ms_per_step": round(sec * 1e3, 3), "value": round(closest / sec / 1e6, 2), "unit": "Mrays/s", "steps": frames, "frames_in_flight": 1,
               "kernel_ms": round(float(np.mean(k_ms)), 3),
               "rays": {"closest": closest, "note": "estimate: reference-algorithm queries of every %d-th row, scaled to the frame" % step}}
        if not args.no_verify:
            step2 = H2 // 3
            n2 = (H2 + step2 - 1) // step2
            w2 = oracle_lib.render_strided(sc2, W2, H2, spp2, depth2, seed, row0=0, rows=n2, row_step=step2, threads=host_threads())[0]
            bad2 = int((out2[::step2].cpu().numpy() != w2).any(axis=-1).sum())
            leg["verified"] = {"rows": n2, "row_step": step2, "pixels": int(n2 * W2), "differing_pixels": bad2, "identical": bad2 == 0}
        pm, pm_path, pm_stale = load_pmc(name)
        if pm and not pm_stale:
            pf = pm["per_frame"]
            leg["roofline"] = {"pmc": pm_path, "valu_issue_frac": round(pf["valu_wave_instructions"] / (leg["kernel_ms"] * 1e-3) / 1e9 / (256 * 4 * 2.4 / 2.0), 4),
                               "lane_utilisation": round(pf["lane_utilisation"], 4), "traffic": int(pf["hbm_bytes"])}
        rt_solo.upload_scene(scene)
        return leg

    if rank == 0:
        rays = tot["closest"]                       # primary + secondary closest-hit queries (renderer.h:40)
        sec_per_step = elapsed / args.steps
        value = rays / sec_per_step / 1e6
        whole_region = flops / sec_per_step / 1e12 / world     # lower bound: everything of a frame, host gaps included
        algorithmic = flops / (kms * 1e-3) / 1e12 / world      # per GPU: slowest rank's kernel time
        basis = ("the span of the frame's render kernels (direct, listed, heavy: they overlap) from HIP events on the context's stream" +
                 ("" if in_flight == 1 else ", measured on three frames run one at a time after the timed region (inside it "
                  "the kernels of the %d frames in flight overlap)" % in_flight))
        # What the kernels execute, from the committed PMC passes of this command (counts per frame do not depend on the
        # clock) over the kernel time measured here.  Peak issue rate: one wave64 VALU instruction per SIMD per 2 cycles,
        # 1024 SIMDs x 2.4 GHz / 2 -- the rate the 157.3 TFLOP/s figure assumes with every slot an FMA.
        peak_issue = 256 * 4 * 2.4e9 / 2.0 / 1e9                 # G wave instructions / s
        pmc, pmc_path, pmc_stale = load_pmc(args.workload)
        roof = {"bound": "valu", "achieved": None, "peak": round(peak_issue, 1), "unit": "G wave64 VALU instructions/s", "frac": None,
                "traffic": None, "kernel_ms": round(kms, 3), "pmc": pmc_path, "pmc_stale": bool(pmc_stale),
                "kernel_source_sha256": kernel_source_hash()}
        if pmc_stale:
            roof["stale_note"] = ("%s was counted on other kernel sources than this tree's (its hash: %s): frac / achieved / traffic are "
                                  "left out rather than priced with it; rerun tools/profile_gpu.sh + tools/pmc_frame.py" % (pmc_path, pmc.get("kernel_source_sha256")))
        if pmc and world == 1 and not pmc_stale:
            pf = pmc["per_frame"]
            issue = pf["valu_wave_instructions"] / (kms * 1e-3) / 1e9
            ex = pf["executed_flops"] / (kms * 1e-3) / 1e12
            roof.update({"achieved": round(issue, 1), "frac": round(issue / peak_issue, 4), "valu_issue_frac": round(issue / peak_issue, 4),
                         "executed_flops": pf["executed_flops"], "executed_tflops": round(ex, 2), "executed_frac": round(ex / PEAK_FP32_TFLOPS, 4),
                         "timed_region_valu_issue_frac": round(pf["valu_wave_instructions"] / sec_per_step / 1e9 / peak_issue, 4),
                         "fma_share": round(pf["fma_share_of_fp32_instructions"], 4), "lane_utilisation": round(pf["lane_utilisation"], 4),
                         "traffic": int(pf["hbm_bytes"]), "traffic_note": "HBM bytes of one frame, 2 x FETCH_SIZE + WRITE_SIZE over its kernels; "
                         "algorithmic bytes: %d (3 B per pixel)" % (3 * W * H), "pmc": pmc_path})
        roof.update({"algorithmic_flops": flops, "algorithmic_tflops": round(algorithmic, 3), "algorithmic_frac": round(algorithmic / PEAK_FP32_TFLOPS, 4),
                     "peak_fp32_tflops": PEAK_FP32_TFLOPS, "timed_region_algorithmic_frac": round(whole_region / PEAK_FP32_TFLOPS, 4),
                     "note": "frac = VALU issue fraction of a frame: executed wave64 VALU instructions (rocprofv3 --pmc SQ_INSTS_VALU, "
                             "profiles/) / kernel_ms / peak issue rate; kernel_ms = " + basis + ". timed_region_valu_issue_frac = the "
                             "same instructions / ms_per_step (the timed region as a whole, host gaps included). executed_* = FP32 operations the "
                             "kernels execute (add + mul + trans + 2 x fma wave instructions x 64 x lane utilisation): nothing is "
                             "contracted (bit parity), so 0.5 of the FMA peak is the ceiling. algorithmic_* = SURVEY 8d flop table x the "
                             "reference algorithm's query tallies: work the kernels partly avoid (candidate lists, shadow pruning), "
                             "not hardware utilisation."})
        line = {
            "metric": "Mrays/s (primary+secondary) at 4K/64spp; % FP32 VALU peak" + (
                "" if in_flight == 1 else " [throughput of a sequence of frames, %d in flight]" % in_flight),
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H, "spp": samples, "max_depth": depth,
                       "seed_base": seed, "parallelism": "cyclic %d-row bands over %d GPU(s), gather to rank 0: %s" % (
                           BAND_ROWS, world, "none (one rank)" if world == 1 else "rt_gather_rows (RCCL send/recv group)" if comm else "torch.distributed.gather (%s; %s)" % (backend, gather_mode)),
                       "frames_in_flight": in_flight, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "host_waits": host_waits},
            "frame_latency_ms": round(float(np.median(latency_ms)), 3),
            "rays": {"primary": tot["primary"], "secondary": tot["closest"] - tot["primary"], "shadow": tot["shadow"],
                     "primitive_tests": tot["tests"], "rng_draws": tot["rand_draws"],
                     "note": tally_note},
            "verified": verified,
            "kernel_ms": round(kms, 3), "device_ms_rank0": round(float(np.mean(device_ms)), 3),
            # where a multi-GPU frame's time goes: RCCL's own count of the communicator's ranks (None: no RCCL communicator
            # -- one rank, or the torch.distributed fallback), the ranks' kernel spans, rank 0's gather + reassembly step per
            # frame (device time of rt_gather_rows' RCCL group and rt_deinterleave_bands; wall time on the fallback path)
            "multi_gpu": {"ranks": world, "rccl_ranks": ra.Renderer.comm_count(comm[0]) if comm else None,
                          "gather_path": "none (one rank): rt_deinterleave_bands only" if world == 1 else "rt_gather_rows" if comm else "torch.distributed.gather (%s)" % backend,
                          "kernel_ms_min_over_ranks": round(kms_min, 3), "kernel_ms_max_over_ranks": round(kms, 3),
                          "gather_ms_rank0": {"mean": round(float(np.mean(gather_ms)), 4), "max": round(float(np.max(gather_ms)), 4)} if gather_ms else None,
                          "gathered_frame_verified_against_oracle": None if verified is None else verified["identical"]},
            "roofline": roof,
        }
        if world == 1 and not args.no_extras and args.workload == "default4_4k_64spp" and args.flags == 0:
            progress("extra legs: relaxed arithmetic, glass 1080p, 2049 hittables 1080p")
            line["relaxed"] = relaxed_leg()
            line["secondary"] = [glass_leg()]
            progress("glass leg done")
            line["secondary"].append(large_scene_leg())
        if not args.no_cpu_baseline and world == 1:
            def count_rays(step):
                n = (H + step - 1) // step
                ps = rt.params(W, H, samples, depth, seed, row0=0, rows=n, band_rows=1, band_stride=step)
                return rt.render_rows(cam, ps, block[:n], count=True)["closest"]
            progress("timing the CPU baseline")
            line["cpu_baseline"] = cpu_baseline(scene, W, H, samples, depth, count_rays)
        print(json.dumps(line), flush=True)
        if verified is not None and not verified["identical"]:
            raise SystemExit("bench.py: the timed frame differs from the CPU oracle in %d pixels" % verified["differing_pixels"])
        for leg in [line.get("relaxed")] + (line.get("secondary") or []):
            v = (leg or {}).get("verified")
            if v and v.get("identical") is False:
                raise SystemExit("bench.py: %s differs from the CPU oracle in %d pixels" % (leg["workload"], v["differing_pixels"]))
            if v and v.get("within_tolerance") is False:
                raise SystemExit("bench.py: the relaxed-arithmetic frame is outside the north star's tolerance: %r" % (v,))
    if comm:
        consumer.comm_destroy(comm[0])
    consumer.close()
    for r in rts:
        r.close()
    if rt_solo is not rt:
        rt_solo.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
