#!/bin/bash
# C4 A/B of experiment builds: per library occupancy / VALU counters, HBM traffic and the timing loop
for s in "$@"; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$s.so
  echo "== $s"
  tools/pmc_quick.sh c4ab_$s many2048_1080p_32spp 0 | head -2 | cut -c1-40,70-400
  tools/pmc_traffic.sh c4ab_$s many2048_1080p_32spp 0 | head -1 | cut -c1-40,70-400
  timeout -k 10 120 python tools/time_workload.py many2048_1080p_32spp 0 | tail -n 2
done
