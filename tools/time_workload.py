#!/usr/bin/env python3
"""Times rt_render_rows for one workload of bench.py (or scene:W:H:spp:depth) without the tally pass:
tools/time_workload.py <workload> [flags...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from bench import WORKLOADS

wl = sys.argv[1]
flags = [int(f) for f in sys.argv[2:]] or [0]
if ":" in wl:      # scene:W:H:spp:depth
    scene_name, W, H, samples, depth = wl.split(":")
    W, H, samples, depth = int(W), int(H), int(samples), int(depth)
else:
    scene_name, W, H, samples, depth = WORKLOADS[wl]
scene = ra.scenes.BUILTIN[scene_name]()
rt = ra.Renderer(0)
rt.upload_scene(scene)
cam = ra.camera35f(scene, W, H)
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ranks = int(os.environ.get("RT_EMULATE_RANKS", "1"))       # time rank 0's share of an N-GPU frame (cyclic 16-row bands)
for f in flags:
    if ranks > 1:
        from raytracer_amd import bands
        p = bands.band_params(rt, W, H, samples, depth, 0, 16, ranks, 0, flags=f)
    else:
        p = rt.params(W, H, samples, depth, 0, flags=f)
    for rep in range(3):
        t0 = time.perf_counter()
        rt.render_rows(cam, p, out)
        print("%s flags %d: wall %.1f ms, device %.1f ms, checksum %d" % (wl, f, (time.perf_counter() - t0) * 1e3, rt.last_device_ms,
                                                                          int(out.sum(dtype=torch.int64))), flush=True)
