#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into profiles/<tag>_summary.md + the raw kernel stats CSV.
usage: tools/summarize_profile.py <tag> [note...]"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
note = " ".join(sys.argv[2:])
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))

# the timed kernels = rt:: kernels launched once per step (the tally build runs once); dominant = largest total time
timed = [r for r in stats if ("rt::" in r["Name"] or "rt_relaxed::" in r["Name"]) and int(r["Calls"]) >= 2
         and any(k in r["Name"] for k in ("render_kernel", "direct_kernel", "heavy_kernel", "classify_kernel", "coop_kernel"))]
# the direct kernel carries ~80 % of a frame's VALU instructions: it leads (large scenes: the cooperative kernel
# does); then by total time
lead = "coop_kernel" if any("coop_kernel" in r["Name"] for r in timed) else "direct_kernel"
timed.sort(key=lambda r: (0 if lead in r["Name"] else 1, -float(r["TotalDurationNs"])))

lines = ["# rocprofv3 summary `%s`" % tag, "", note, "",
         "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline " + os.environ.get("RT_PROFILE_ARGS", "--frames-in-flight 1") + "` (frames one at a time: with frames in flight the kernels of different frames overlap and a launch's duration says nothing about it) and one "
         "`rocprofv3 --pmc ...` run per counter group (tools/profile_gpu.sh).  Under `--pmc` rocprofv3 serialises the "
         "dispatches, so the counters below are per kernel even though the kernels of a frame overlap in a normal run.", "",
         "## Kernel trace (`%s_kernel_stats.csv`)" % tag, "", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
for r in stats[:9]:
    lines.append("| `%s` | %s | %.3f | %s |" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) * 1e-6, r["Percentage"]))


def kernel_section(main, title):
    name = main["Name"]
    agg = collections.defaultdict(float)
    calls = collections.Counter()
    meta = {}
    durs = []
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_*_counter_collection.csv"))):
        seen = set()
        rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"] == name]
        for r in rows:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
        for c in {r["Counter_Name"] for r in rows}:
            calls[c] = len(seen)
    per = {k: v / max(calls[k], 1) for k, v in agg.items()}       # per launch
    avg_ms = float(main["AverageNs"]) * 1e-6
    out = ["", "## %s: `%s`" % (title, name), "", "avg %.3f ms in the kernel trace (overlapping the frame's other kernels); launch: " % avg_ms
           + ", ".join("%s=%s" % kv for kv in meta.items()), "", "| counter (per launch) | value |", "|---|---|"]
    for k in sorted(per):
        out.append("| %s | %.6g |" % (k, per[k]))
    g = per.get
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
        cycles = g("GRBM_GUI_ACTIVE") / 8.0                      # summed over the 8 XCDs; the kernel alone on the GPU
        simds = 256 * 4
        issue = g("SQ_INSTS_VALU") / (simds * cycles / 2.0)      # a wave64 VALU instruction occupies a SIMD-32 for 2 cycles
        out += ["", "Derived (serialised run: GRBM_GUI_ACTIVE / 8 = %.3g cycles = %.3f ms at 2.4 GHz):" % (cycles, cycles / 2.4e6), "",
                "* VALU issue utilisation = SQ_INSTS_VALU / (1024 SIMDs x cycles / 2) = **%.1f %%** of the one-wave64-VALU-per-2-cycles peak" % (100 * issue),
                "* VALU instructions per wave = %.0f; SALU per VALU = %.2f; LDS per VALU = %.3f" % (
                    g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_SALU", 0) / g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS", 0) / g("SQ_INSTS_VALU"))]
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            out.append("* lane utilisation (branch divergence) = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = **%.1f %%**" % (
                100 * g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))))
        if g("SQ_WAVE_CYCLES"):
            out.append("* mean resident waves per SIMD = 4 x SQ_WAVE_CYCLES / cycles / 1024 = **%.2f**" % (4 * g("SQ_WAVE_CYCLES") / cycles / simds))
        if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_ANY" in per:
            out.append("* wave time split: issuing %.0f %%, waiting on s_waitcnt/barrier (SQ_WAIT_ANY) %.0f %%, issue-stalled (SQ_WAIT_INST_ANY) %.0f %%" % (
                100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
        arith = sum(g(k, 0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"))
        if arith:
            out.append("* FP32 arithmetic share of VALU instructions (add+mul+fma+trans) = %.0f %%; int32 %.0f %%" % (
                100 * arith / g("SQ_INSTS_VALU"), 100 * g("SQ_INSTS_VALU_INT32", 0) / g("SQ_INSTS_VALU")))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        out.append("* HBM traffic per launch: FETCH_SIZE %.1f KB (x2 per the gfx950 correction = %.1f KB), WRITE_SIZE %.1f KB -- the path is not "
                   "HBM-bound (framebuffer = 3 B/pixel)" % (g("FETCH_SIZE"), 2 * g("FETCH_SIZE"), g("WRITE_SIZE")))
    return out


for n, r in enumerate(timed[:5]):
    lines += kernel_section(r, "Dominant kernel" if n == 0 else "Kernel")
with open(os.path.join(dst, tag + "_summary.md"), "w") as fh:
    fh.write("\n".join(lines) + "\n")
print("\n".join(lines))
