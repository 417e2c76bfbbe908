#!/bin/bash
# kernel timeline of a 20-frame run of rank 0's share of an 8-GPU frame, 16 in flight (what bench.py --steps 20 times at N = 8)
export RT_EMULATE_RANKS=8 RT_HW_QUEUES=16 TMPDIR=/tmp
OUT=gpurun_out/trace_short; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/pipeline_time.py default4_4k_64spp 20 16 > $OUT/log.txt 2>&1 || { tail $OUT/log.txt; exit 1; }
grep "ms/frame" $OUT/log.txt
python3 tools/short_run_timeline.py $(find $OUT -name "t_kernel_trace.csv" | head -n 1) 20
