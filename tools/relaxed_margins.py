#!/usr/bin/env python3
"""Measured margins of RT_FLAG_RELAXED against the CPU oracle on BASELINE.json's configurations (the cases and the
tolerance of tests/test_gpu_relaxed.py): tools/relaxed_margins.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import raytracer_amd as ra
import test_gpu_relaxed as t

out = {"tolerance": {"lsb": t.LSB_TOLERANCE, "min_within_frac": t.MIN_WITHIN_FRAC, "min_psnr_db": t.MIN_PSNR_DB}, "cases": {}}
with ra.Renderer(0) as rt:
    for case in sorted(t.CASES):
        sc, got, (W, H, samples, depth, row0, n, step) = t.render_case(rt, case, t.RT_FLAG_RELAXED)
        want, _, _ = oracle_lib.render_strided(sc, W, H, samples, depth, 0, row0=row0, rows=n, row_step=step, threads=t.THREADS)
        m = t.margins(got, want)
        m["rows"] = "%d rows from row %d, every %d" % (n, row0, step)
        out["cases"][case] = m
        print(case, json.dumps(m), flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
