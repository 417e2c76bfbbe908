#!/bin/bash
# kernel-trace only: tools/trace_gpu.sh <tag> [bench args]  -> prints per-kernel stats
TAG=$1; shift
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
cut -d, -f1-4,6,7 $OUT/t_kernel_stats.csv | cut -c1-170
