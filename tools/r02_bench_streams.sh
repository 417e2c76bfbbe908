#!/bin/bash
# bench.py with the listed kernels on one stream / two, frames in flight 8 / 16
export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_one.so
run() { echo "== streams $1 queues $2 F=$3"; RT_HW_QUEUES=$2 RT_SEQ_STREAMS=$1 timeout -k 10 200 python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-verify --frames-in-flight $3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['frame_latency_ms'])" || exit 1; }
run 2 16 8; run 1 16 8; run 1 16 16; run 1 20 16; run 1 24 16; run 1 16 12; run 2 16 8; run 1 16 16
