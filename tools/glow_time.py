#!/usr/bin/env python3
"""Times rt_apply_glow on device buffers: tools/glow_time.py [W H spp]   (default 3840 2160 4)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
W, H, spp = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (3840, 2160, 4)))
sc = ra.scenes.default4()
rt = ra.Renderer(0); rt.upload_scene(sc)
cam = ra.camera35f(sc, W, H)
rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"); em = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
rt.render_rows(cam, rt.params(W, H, spp, 10, 0), rgb, em)
base = rgb.clone()
for rep in range(4):
    rgb.copy_(base); torch.cuda.synchronize()
    t0 = time.perf_counter(); levels = rt.apply_glow(rgb, em, W, H); dt = time.perf_counter() - t0
    print("glow %dx%d: %d levels, %.2f ms, checksum %d" % (W, H, levels, dt * 1e3, int(rgb.sum(dtype=torch.int64))), flush=True)
