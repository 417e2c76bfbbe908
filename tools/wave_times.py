#!/usr/bin/env python3
"""Experiment (library built with -DRT_WAVE_TIMES): the listed kernel's waves of a sequence of frames in flight -- how long
they live, how many are resident, how they spread over the SIMDs.
    tools/wave_times.py <workload> <frames> <in flight>"""
import ctypes
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as ra
from raytracer_amd import bands
from raytracer_amd.pipeline import FramePipeline
from bench import WORKLOADS

wl, frames, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
scene_name, W, H, samples, depth = WORKLOADS[wl]
scene = ra.scenes.BUILTIN[scene_name]()
cam = ra.camera35f(scene, W, H)
rts = [ra.Renderer(0) for _ in range(F)]
for rt in rts:
    rt.upload_scene(scene)
p = bands.band_params(rts[0], W, H, samples, depth, 0, 16, 1, 0, flags=ra._abi.RT_FLAG_SEQUENCE if F > 1 else 0)
blocks = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2 * F)]
pipe = FramePipeline(lambda w, b, k: rts[w].render_rows(cam, p, b), blocks, H, 16, rts[0], workers=F)
lib = ra._abi.load()
lib.rt_debug_wave_times.restype = ctypes.c_int
cap = 1 << 20
buf = np.zeros((cap, 4), dtype=np.uint32)
pipe.run(F)
lib.rt_debug_wave_times(buf.ctypes.data_as(ctypes.c_void_p), cap)       # drop the warm-up's
import time
t0 = time.perf_counter()
pipe.run(frames)
wall = (time.perf_counter() - t0) * 1e3
n = lib.rt_debug_wave_times(buf.ctypes.data_as(ctypes.c_void_p), cap)
r = buf[:n]
start = (r[:, 0] - r[:, 0].min()).astype(np.int64) / 1e5            # ms (100 MHz)
dur = r[:, 1].astype(np.int64) / 1e5
hw = r[:, 2]
xcc = r[:, 3] & 15
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
unit = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print("%s: %d frames, %d in flight, %.2f ms/frame; %d waves (%.0f per frame)" % (wl, frames, F, wall / frames, n, n / frames))
print("wave lifetime ms: mean %.2f, median %.2f, p90 %.2f, p99 %.2f, max %.2f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("distinct SIMDs seen: %d; xcc %s se %s sh %s cu %s" % (len(np.unique(unit)), np.unique(xcc), np.unique(se), np.unique(sh), np.unique(cu)))
span = (start + dur).max()
print("span %.1f ms; mean resident listed waves %.0f (%.2f per SIMD of 1024)" % (span, dur.sum() / span, dur.sum() / span / 1024))
# residency over time, and how unevenly: per 10 ms, waves resident in total / on the busiest SIMD / SIMDs with none
edges = np.arange(0, span, span / 24)
for t in edges:
    live = (start <= t) & (start + dur > t)
    per = np.bincount(unit[live], minlength=int(unit.max()) + 1)
    seen = per[np.unique(unit)]
    print("  t=%7.1f ms: %5d waves resident; per SIMD mean %.2f max %d; SIMDs idle %4d; lifetime so far of the residents: mean %.1f ms" % (
        t, live.sum(), seen.mean(), seen.max(), (seen == 0).sum(), (t - start[live]).mean() if live.any() else 0.0))
tot = np.bincount(unit, weights=dur, minlength=int(unit.max()) + 1)[np.unique(unit)]
cnt = np.bincount(unit, minlength=int(unit.max()) + 1)[np.unique(unit)]
print("per SIMD over the run: waves mean %.1f min %d max %d; wave-ms mean %.1f min %.1f max %.1f" % (cnt.mean(), cnt.min(), cnt.max(), tot.mean(), tot.min(), tot.max()))
# how many listed waves a wave shares its SIMD with, at its start
order = np.argsort(start)
comp = np.zeros(n)
for u in np.unique(unit):
    m = np.where(unit == u)[0]
    st, en = start[m], start[m] + dur[m]
    comp[m] = [((st <= st[i]) & (en > st[i])).sum() for i in range(len(m))]
for c in range(1, 9):
    sel = comp == c
    if sel.any():
        print("  waves that start as number %d on their SIMD: %6d, lifetime mean %.2f ms, p99 %.2f, max %.2f" % (c, sel.sum(), dur[sel].mean(), np.percentile(dur[sel], 99), dur[sel].max()))
# does a wave run slower in company?  speed = this wave's lifetime against the same wave (workgroup, wave order unknown: workgroup's longest) alone
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wave_times_%s_%d.npy" % (wl, F)), r)
