#!/usr/bin/env python3
"""Timing experiment: DEFAULT-4 with individual spheres' materials replaced by the diffuse one (which bounce costs what)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as ra

W, H, S, D = 3840, 2160, 64, 10
rt = ra.Renderer(0)
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
cam = ra.camera35f(60.0, W, H)
for name, repl in [("default4", {}), ("glass->diffuse", {3: 0}), ("mirror->diffuse", {2: 0}), ("red->diffuse", {1: 0}),
                   ("only mirror", {1: 0, 3: 0}), ("only red", {2: 0, 3: 0}), ("only glass", {1: 0, 2: 0}),
                   ("all diffuse", {1: 0, 2: 0, 3: 0})]:
    sc = ra.scenes.default4()
    for i, m in repl.items():
        h = sc.hittables[i]
        sc.hittables[i] = (h[0], h[1], h[2], h[3], m)
    rt.upload_scene(sc)
    ranks = int(os.environ.get("RT_EMULATE_RANKS", "1"))      # rank 0's share of an N-GPU frame
    if ranks > 1:
        from raytracer_amd import bands
        p = bands.band_params(rt, W, H, S, D, 0, 16, ranks, 0)
    else:
        p = rt.params(W, H, S, D, 0)
    ts = []
    for _ in range(3):
        rt.render_rows(cam, p, out)
        ts.append(rt.last_kernel_ms)
    print("%-16s kernel %.2f ms" % (name, min(ts)), flush=True)
