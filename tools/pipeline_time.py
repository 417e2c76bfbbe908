#!/usr/bin/env python3
"""Times a sequence of frames with F frames in flight (raytracer_amd/pipeline.py) on one GPU:
    tools/pipeline_time.py <workload> <frames> <F> [<F> ...]        F = producers (contexts)
RT_EMULATE_RANKS=N renders rank 0's share of an N-GPU frame (cyclic 16-row bands), i.e. what one rank of an N-GPU
job does per frame minus the RCCL gather.  RT_HW_QUEUES sets GPU_MAX_HW_QUEUES before HIP starts; RT_FLAGS = rt_params.flags
(include/rt_amd.h) for A/B runs; RT_CHECK_ALL=1 checksums every frame (slower: a check, not a timing)."""
import os
import sys
import time

if os.environ.get("RT_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["RT_HW_QUEUES"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from raytracer_amd import bands
from raytracer_amd.pipeline import FramePipeline
from bench import WORKLOADS

wl, frames = sys.argv[1], int(sys.argv[2])
scene_name, W, H, samples, depth = WORKLOADS[wl]
depth = int(os.environ.get("RT_DEPTH", depth))          # (0: no pixel bounces -- the direct kernel alone, no classification)
scene = ra.scenes.BUILTIN[scene_name]()
ranks = int(os.environ.get("RT_EMULATE_RANKS", "1"))
cam = ra.camera35f(scene, W, H)
rpr = bands.rows_per_rank(H, 16, ranks)
ref = None
for F in [int(f) for f in sys.argv[3:]] or [1]:
    rts = [ra.Renderer(0) for _ in range(F)]
    for rt in rts:
        rt.upload_scene(scene)
    p = bands.band_params(rts[0], W, H, samples, depth, 0, 16, ranks, 0, flags=int(os.environ.get("RT_FLAGS", str(ra._abi.RT_FLAG_SEQUENCE if F > 1 else 0))))
    nblocks = F * int(os.environ.get("RT_BLOCKS_PER_WORKER", "2"))
    blocks = [torch.zeros((rpr, W, 3), dtype=torch.uint8, device="cuda") for _ in range(nblocks)]
    torch.cuda.synchronize()
    calls = []

    def render(w, block, k):
        t1 = time.perf_counter()
        rts[w].render_rows(cam, p, block)
        calls.append((t1, time.perf_counter(), w, k))

    pipe = FramePipeline(render, blocks, rpr, 16, rts[0], workers=F)
    pipe.run(F)                                         # warm every context
    for rep in range(2):
        del calls[:]
        sums = []
        check = (lambda k, f: sums.append(int(f.sum(dtype=torch.int64)))) if os.environ.get("RT_CHECK_ALL") else None
        t0 = time.perf_counter()
        last = pipe.run(frames, on_frame=check)
        dt = time.perf_counter() - t0
        if check:       # every frame of the run is the same frame: one checksum
            print("  all %d frames checked: %s" % (len(sums), "one checksum" if len(set(sums)) == 1 else "MISMATCH %s" % sorted(set(sums))[:4]))
        if os.environ.get("RT_SLOW_CALLS"):         # render calls that took longer than RT_SLOW_CALLS ms: start, duration, worker, frame
            for a, b, w, k in sorted(calls):
                if (b - a) * 1e3 > float(os.environ["RT_SLOW_CALLS"]):
                    print("  slow call: start %.2f ms, %.2f ms long, worker %d, frame %d" % ((a - t0) * 1e3, (b - a) * 1e3, w, k))
        cs = int(last.sum(dtype=torch.int64))
        ref = cs if ref is None else ref
        print("%s ranks %d in-flight %d: %.3f ms/frame over %d frames, checksum %d%s" % (
            wl, ranks, F, dt / frames * 1e3, frames, cs, "" if cs == ref else "  MISMATCH"), flush=True)
    for rt in rts:
        rt.close()
