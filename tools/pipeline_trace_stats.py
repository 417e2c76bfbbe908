#!/usr/bin/env python3
"""Per-kernel durations and the busy time of the GPU from a rocprofv3 kernel trace (csv): how much of the wall time
has at least one kernel running, and how long each kernel takes when frames overlap."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per = defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")
    per[name].append((e - s) / 1e3)
    ev.append((s, e))
ev.sort()
# the last third of the trace: the steady state of the last run
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 2 // 3
busy, cur_s, cur_e = 0, None, None
for s, e in ev:
    if e < t_lo:
        continue
    s = max(s, t_lo)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
span = ev[-1][1] - t_lo
print("last third of the trace: %.2f ms, some kernel running %.1f %% of it" % (span / 1e6, 100.0 * busy / span))
for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    d2 = sorted(d)
    print("%-28s n %5d  mean %9.1f us  median %9.1f  max %9.1f  total %9.2f ms" % (name[:28], len(d), sum(d) / len(d), d2[len(d) // 2], d2[-1], sum(d) / 1e3))

# how many direct kernels run at once, over the middle of the last run (between its 3rd and 3rd-last direct kernel)
d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "direct_kernel" in r["Kernel_Name"])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = d[-nlast:]
lo, hi = d[2][0], d[-3][1]
pts = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < lo or s > hi:
        continue
    kind = "direct" if "direct_kernel" in r["Kernel_Name"] else ("listed" if ("render_kernel" in r["Kernel_Name"] or "heavy" in r["Kernel_Name"]) else "other")
    pts.append((max(s, lo), 1, kind)); pts.append((min(e, hi), -1, kind))
pts.sort()
cur = defaultdict(int); last = lo; hist = defaultdict(float); tw = defaultdict(float)
for t, dv, kind in pts:
    hist[cur["direct"]] += t - last
    for k2, v in cur.items():
        tw[k2] += v * (t - last)
    cur[kind] += dv; last = t
tot = float(hi - lo)
print("steady window %.2f ms = %.3f ms per frame" % (tot / 1e6, tot / 1e6 / (len(d) - 5)))
print("direct kernels running at once: " + ", ".join("%d: %.1f %%" % (k2, 100 * v / tot) for k2, v in sorted(hist.items())))
print("mean kernels in flight: " + ", ".join("%s %.2f" % (k2, v / tot) for k2, v in sorted(tw.items())))
