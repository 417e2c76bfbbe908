#!/usr/bin/env python3
"""Large-scene timing probe: many_objects as it is (spheres + cubes: type-sorted dealing through `order`) against the
same layout with spheres only (no indirection): tools/coop_probe.py [W H spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as ra
from raytracer_amd import scenes

W, H, spp = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (480, 270, 32)
mixed = scenes.many_objects()
only = scenes.Scene(name="spheres_only")
only.materials = list(mixed.materials); only.lights = list(mixed.lights)
only.fov, only.znear, only.zfar = mixed.fov, mixed.znear, mixed.zfar
for h in mixed.hittables:
    if h[0] == 2:       # cube (corner, size) -> sphere of the same footprint
        a, size = np.array(h[1]), np.array(h[2])
        r = float(size[0] / 2)
        only.sphere((float(a[0] + r), r, float(a[2] + r)), r, h[4])
    else:
        only.hittables.append(h)
rt = ra.Renderer(0)
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for name, sc in (("mixed", mixed), ("spheres", only)):
    rt.upload_scene(sc)
    cam = ra.camera35f(sc, W, H)
    p = rt.params(W, H, spp, 10, 0)
    cnt = {} if os.environ.get("RT_NO_TALLY") else rt.render_rows(ra.camera35f(sc, 96, 54), rt.params(96, 54, spp, 10, 0), out[:54, :96].contiguous(), count=True)
    print(name, "tally at 96x54:", {k: "%.3g" % v for k, v in cnt.items()}, flush=True)
    for rep in range(2):
        t0 = time.perf_counter(); rt.render_rows(cam, p, out); dt = time.perf_counter() - t0
        print("%s %d hittables: %.1f ms (device %.1f)" % (name, len(sc.hittables), dt * 1e3, rt.last_device_ms), flush=True)
