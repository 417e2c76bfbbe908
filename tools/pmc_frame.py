#!/usr/bin/env python3
"""Per-frame totals of the PMC passes of tools/profile_gpu.sh: gpurun_out/prof_<tag>/pmc_*/ -> profiles/pmc_<workload>.json,
the file bench.py prices its roofline object with (executed VALU instructions, executed FP32 operations, HBM bytes of one
frame; they do not depend on the clock, bench.py divides them by the kernel time it measures itself).
usage: tools/pmc_frame.py <tag> <workload>

Per kernel and launch (rocprofv3 serialises dispatches under --pmc, so counters are per kernel): the counters' sums over
the launches / the number of launches.  A frame = every rt:: kernel launched at least twice in the run (the tally build
runs once and is left out), each weighted by launches / frames, frames = launches of the direct kernel."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rt::" not in k and "rt_relaxed::" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
kernels = {}
for k, cs in agg.items():
    n = max(len(v) for v in launches[k].values())
    if n < 2:
        continue
    kernels[k] = {"launches": n, "per_launch": {c: v / len(launches[k][c]) for c, v in cs.items()}}
frames = max((v["launches"] for k, v in kernels.items() if "direct_kernel" in k), default=0)
if not frames:
    raise SystemExit("no direct_kernel launches found under " + src)
tot = collections.defaultdict(float)
for k, v in kernels.items():
    w = v["launches"] / frames
    v["launches_per_frame"] = w
    for c, x in v["per_launch"].items():
        tot[c] += w * x
g = tot.get
lane_util = g("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * g("SQ_ACTIVE_INST_VALU", 1))
fp = {c: g("SQ_INSTS_VALU_" + c + "_F32", 0.0) for c in ("ADD", "MUL", "FMA", "TRANS")}
sys.path.insert(0, ROOT)
from bench import kernel_source_hash
out = {
    "workload": workload, "profile": "prof_" + tag, "frames_in_run": frames,
    # the kernels these counts belong to: bench.py prices a run with them only if its tree hashes the same
    "kernel_source_sha256": kernel_source_hash(),
    "per_frame": {
        "valu_wave_instructions": g("SQ_INSTS_VALU", 0),
        "salu_wave_instructions": g("SQ_INSTS_SALU", 0),
        "lane_utilisation": lane_util,
        "fp32_wave_instructions": {k.lower(): v for k, v in fp.items()},
        # executed FP32 operations: wave instructions x 64 lanes x the share of lanes that are on (an FMA counts 2)
        "executed_flops": (fp["ADD"] + fp["MUL"] + fp["TRANS"] + 2.0 * fp["FMA"]) * 64.0 * lane_util,
        "fma_share_of_fp32_instructions": fp["FMA"] / max(sum(fp.values()), 1.0),
        # MI355X_MICROARCH.md: FETCH_SIZE counts 32 B units reported in KB and under-reports by 2 on gfx950; WRITE_SIZE in KB
        "hbm_bytes": (2.0 * g("FETCH_SIZE", 0) + g("WRITE_SIZE", 0)) * 1024.0,
        "fetch_kb": g("FETCH_SIZE", 0), "write_kb": g("WRITE_SIZE", 0),
    },
    "kernels": {k: {"launches_per_frame": v["launches_per_frame"], "valu_wave_instructions": v["per_launch"].get("SQ_INSTS_VALU", 0),
                    "gui_active_cycles": v["per_launch"].get("GRBM_GUI_ACTIVE", 0) / 8.0} for k, v in kernels.items()},
}
dst = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)
with open(dst, "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(json.dumps(out["per_frame"], indent=1))
