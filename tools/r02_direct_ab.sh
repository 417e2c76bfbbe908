#!/bin/bash
# direct-kernel variants (tools/devbuild.sh <suffix> -D...): serialised counters, a frame alone, frames in flight
set -e
for v in "$@"; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$v.so
  echo "== $v"
  bash tools/pmc_quick.sh d_$v default4_4k_64spp 0 | grep direct_kernel
  timeout -k 10 120 python tools/time_workload.py default4_4k_64spp 0 2>&1 | grep -v amdgpu.ids | tail -1
  RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 24 8 2>&1 | grep -v amdgpu.ids | tail -1
  timeout -k 10 120 python tools/time_workload.py glass_1080p_64spp 0 2>&1 | grep -v amdgpu.ids | tail -1
done
