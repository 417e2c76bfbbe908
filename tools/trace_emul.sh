#!/bin/bash
# kernel trace of rank 0's share of an N-GPU frame: tools/trace_emul.sh <tag> <ranks> <lib suffix|main>
TAG=$1; export RT_EMULATE_RANKS=$2; v=$3
lib=raytracer_amd/librt_amd_$v.so; [ "$v" = "main" ] && lib=raytracer_amd/librt_amd.so
export RT_AMD_LIB=$PWD/$lib TMPDIR=/tmp
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/time_workload.py default4_4k_64spp 0 > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
