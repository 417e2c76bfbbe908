#!/bin/bash
# kernel trace (durations per kernel) of one workload: tools/r02_trace.sh <tag> <workload> [flags]
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/kt_$TAG; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/time_workload.py "$@" > $OUT/run.log 2>&1 || exit 1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows[:12]:
    print("%-70s calls %4s avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
