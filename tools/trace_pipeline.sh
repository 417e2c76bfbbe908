#!/bin/bash
# kernel trace of frames in flight: tools/trace_pipeline.sh <tag> <ranks> <F> [frames] [queues]
TAG=$1; export RT_EMULATE_RANKS=$2 RT_HW_QUEUES=${5:-16} TMPDIR=/tmp
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/pipeline_time.py default4_4k_64spp ${4:-30} $3 > $OUT/log.txt 2>&1
grep "ms/frame" $OUT/log.txt | tail -3
python3 tools/pipeline_trace_stats.py $OUT/t_kernel_trace.csv ${4:-30}
