#!/bin/bash
# per-frame kernel durations and start offsets of C4 for an experiment build: tools/c4_trace_reps.sh <suffix>
export TMPDIR=/tmp
S=$1; OUT=gpurun_out/c4tr_$S; mkdir -p $OUT
RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$S.so timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/time_workload.py many2048_1080p_32spp 0 0 > $OUT/run.log 2>&1 || exit 1
grep wall $OUT/run.log | awk '{print $5}' | tr "\n" " "; echo
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")) if "rt::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
for r in rows:
    n=r["Kernel_Name"]
    if "classify" in n: t0=int(r["Start_Timestamp"]); print("--")
    if "coop" in n or "direct" in n:
        print("%-28s start %8.2f dur %8.2f ms  lds %s scratch %s vgpr %s" % (n[9:37], (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("VGPR_Count")))
PY
