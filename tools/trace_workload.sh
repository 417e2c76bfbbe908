#!/bin/bash
# kernel trace of tools/time_workload.py: tools/trace_workload.sh <tag> <workload or scene:W:H:spp:depth> [flags...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/time_workload.py "$@" > $OUT/log.txt 2>&1
grep "wall" $OUT/log.txt | tail -3
cut -d, -f1-4 $OUT/t_kernel_stats.csv | cut -c1-150 | head -8
