#!/bin/bash
# kernel timeline of the last frame of tools/time_workload.py: tools/r02_timeline.sh <tag> <workload> [flags]
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/tl_$TAG; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/time_workload.py "$@" > $OUT/run.log 2>&1 || exit 1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")) if "rt::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last frame = from the last classify kernel on
last=max(i for i,r in enumerate(rows) if "classify" in r["Kernel_Name"])
t0=int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%-60s start %9.3f ms  dur %9.3f ms  grid %s" % (r["Kernel_Name"][9:69], (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
