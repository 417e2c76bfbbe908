import sys, time, numpy as np
sys.path.insert(0, '.')
import raytracer_amd as ra
sc = ra.scenes.default4(); rt = ra.Renderer(0); rt.upload_scene(sc)
W, H = 3840, 2160
cam = ra.camera35f(sc, W, H); p = rt.params(W, H, 64, 10, 0)
rgb = np.zeros((H, W, 3), np.uint8)
for i in range(4):
    t = time.perf_counter(); rt.render_rows(cam, p, rgb); dt = (time.perf_counter() - t) * 1e3
    print("host buffer: wall %.2f ms, device %.2f ms" % (dt, rt.last_device_ms))
