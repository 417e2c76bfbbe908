#!/bin/bash
# direct-kernel A/B: counters, traffic of the direct kernel, two in-flight timings and the frame alone
for s in "$@"; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$s.so
  echo "== $s"
  tools/pmc_quick.sh ab_$s default4_4k_64spp 0 | head -2 | cut -c1-60,90-400
  tools/pmc_traffic.sh ab_$s default4_4k_64spp 0 | grep direct | cut -c1-40,70-300
  for i in 1 2 3; do RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 64 16 2>/dev/null | tail -n 1; done
  timeout -k 10 120 python tools/time_workload.py default4_4k_64spp 0 | tail -n 1
done
