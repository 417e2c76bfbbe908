#!/usr/bin/env python3
"""Full-size goldens from the REAL reference (oracle/_ref/ref_oracle, single thread, time() pinned): BASELINE.json configs
C1 (1024x512 / 5 spp / depth 10) and C2 (1920x1080 / 16 spp / depth 5) of DEFAULT-4.  The frames are too large to commit;
committed are the hashes SURVEY.md Appendix C names (FNV-1a32 of the PPM body, sha256 of the PPM file), the query
tallies, and every 32nd row of each frame (tests/golden/fullsize.npz) so that a mismatch can be located.
Runs only in the authoring container:   make -C oracle ref && python tests/golden/make_fullsize.py"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import run_ref, OUT, scenes  # noqa: E402

CASES = [("C1", 1024, 512, 5, 10, 0), ("C2", 1920, 1080, 16, 5, 0)]
ROW_STEP = 32


def fnv1a32(data):
    h = 0x811C9DC5
    for b in memoryview(data):
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h


def main():
    meta, rows = {}, {}
    for name, W, H, samples, depth, seed in CASES:
        sc = scenes.default4()
        info, rgb, em, cam = run_ref(sc, W, H, samples, depth, seed)
        body = rgb.tobytes()
        ppm = b"P6\n%d %d\n255\n" % (W, H) + body
        meta[name] = {"scene": "default4", "width": W, "height": H, "samples": samples, "max_depth": depth, "seed_base": seed,
                      "fnv1a32_body": "%08x" % fnv1a32(body), "sha256_16_ppm": hashlib.sha256(ppm).hexdigest()[:16],
                      "sha256_body": hashlib.sha256(body).hexdigest(),
                      "counters": {k: int(info[k]) for k in ("primary", "closest", "shadow", "tests", "rand_draws")},
                      "row_step": ROW_STEP}
        rows[name + "_rows"] = rgb[::ROW_STEP].copy()
        rows[name + "_emission_rows"] = em[::ROW_STEP].copy()
        print(name, meta[name])
    with open(os.path.join(OUT, "fullsize_hashes.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "fullsize.npz"), **rows)


if __name__ == "__main__":
    main()
