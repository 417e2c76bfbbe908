#!/usr/bin/env python3
"""Timeline of the last short run in a rocprofv3 kernel trace (csv) of tools/pipeline_time.py: per 0.5 ms, how many
direct / listed / other kernels are running (time-weighted means).  tools/short_run_timeline.py <csv> <frames>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2])
ev = []
for r in rows:
    n = r["Kernel_Name"]
    kind = "direct" if "direct_kernel" in n else "listed" if ("render_kernel" in n or "heavy" in n or "resume" in n) else "other"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
ev.sort()
# the last run = the last `frames` direct kernels and everything from 1 ms before the first of them
d = [e for e in ev if e[2] == "direct"][-frames:]
lo = min(s for s, _, _ in d) - 1500000
run = [e for e in ev if e[1] >= lo]
lo = min(s for s, _, _ in run)
hi = max(e for _, e, _ in run)
print("run of %d frames: %.2f ms from the first kernel to the last = %.3f ms per frame" % (frames, (hi - lo) / 1e6, (hi - lo) / 1e6 / frames))
step = 500000
t = lo
while t < hi:
    w = {"direct": 0.0, "listed": 0.0, "other": 0.0}
    for s, e, k in run:
        o = min(e, t + step) - max(s, t)
        if o > 0:
            w[k] += o
    print("%5.1f ms  direct %5.2f  listed %5.2f  other %5.2f" % ((t - lo) / 1e6, w["direct"] / step, w["listed"] / step, w["other"] / step))
    t += step
