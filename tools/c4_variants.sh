#!/bin/bash
# C4 timing over experiment builds of the library: tools/c4_variants.sh <suffix>[:RT_DEBUG_COOP value] ...
for v in "$@"; do
  S=${v%%:*}; D=${v#*:}; [ "$D" = "$v" ] && D=1
  echo "== $S (RT_DEBUG_COOP=$D)"
  RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$S.so RT_DEBUG_COOP=$D timeout -k 10 100 python -u tools/time_workload.py many2048_1080p_32spp 2>&1 | sort | uniq -c | tail -n 6 || exit 1
done
