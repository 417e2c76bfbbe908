#!/bin/bash
# frames in flight: the listed kernels on the context's one stream (RT_SEQ_STREAMS=1) against a second stream (2)
export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_one.so
run() { echo "== ranks $1 streams $2 queues $3 F=$4"; RT_HW_QUEUES=$3 RT_SEQ_STREAMS=$2 RT_EMULATE_RANKS=$1 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 96 $4 2>/dev/null | tail -n 2 || exit 1; }
run 8 2 16 8; run 8 1 16 8; run 8 1 16 12; run 8 1 16 16; run 8 1 16 16; run 8 1 24 24; run 8 1 32 32; run 8 1 16 24
run 4 1 16 16; run 2 1 16 16; run 1 1 16 16; run 1 1 16 8; run 1 2 16 8
