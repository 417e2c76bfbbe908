#!/usr/bin/env python3
"""Condenses gpurun_out/trace_inflight_<workload>/t_kernel_trace.csv (tools/r03_inflight_trace.sh: rocprofv3 --kernel-trace
of a sequence of frames in flight) into profiles/r03_inflight_<workload>.md: per kernel, how long a launch lasts when the
kernels of sixteen frames overlap, and how many run at once.   usage: tools/inflight_trace_summary.py <workload> [note...]"""
import collections, csv, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1]
note = " ".join(sys.argv[2:])
rows = list(csv.DictReader(open(os.path.join(ROOT, "gpurun_out", "trace_inflight_" + wl, "t_kernel_trace.csv"))))
rows = [r for r in rows if "rt::" in r["Kernel_Name"] or "rt_relaxed::" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    n = n[:n.index("(")] if "(" in n else n
    by[n.replace("void ", "")].append(((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r["Queue_Id"]))
span = max(b for v in by.values() for _, b, _ in v)
out = ["# Kernel trace of frames in flight: `%s`" % wl, "", note, "",
       "`rocprofv3 --kernel-trace -- python3 tools/pipeline_time.py %s <frames> 16` with `GPU_MAX_HW_QUEUES=16` (tools/r03_inflight_trace.sh); "
       "%d rt:: launches over %.0f ms, hardware queues seen: %d.  A launch's duration here includes the time its workgroups wait "
       "for room beside the other frames' kernels." % (wl, len(rows), span, len({q for v in by.values() for _, _, q in v})), "",
       "| kernel | launches | mean ms | min ms | max ms | mean running at once |", "|---|---|---|---|---|---|"]
for k, v in sorted(by.items(), key=lambda kv: -sum(b - a for a, b, _ in kv[1])):
    d = [b - a for a, b, _ in v]
    out.append("| `%s` | %d | %.2f | %.2f | %.2f | %.1f |" % (k[:80], len(v), sum(d) / len(d), min(d), max(d), sum(d) / span))
open(os.path.join(ROOT, "profiles", "r03_inflight_%s.md" % wl), "w").write("\n".join(out) + "\n")
print("\n".join(out))
