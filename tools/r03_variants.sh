#!/bin/bash
# library variants (raytracer_amd/variants/librt_amd_<tag>.so, experiments) against the shipped one: frames in flight and alone
#   tools/r03_variants.sh <tag>...      ("main" = the shipped library)
export RT_HW_QUEUES=16
for v in "$@"; do
  if [ "$v" = main ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/raytracer_amd/variants/librt_amd_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/pipeline_time.py glass_1080p_64spp 64 16 2>/dev/null | tail -n 1 || exit 1
  timeout -k 10 200 python tools/pipeline_time.py default4_4k_64spp 64 16 2>/dev/null | tail -n 1 || exit 1
  RT_EMULATE_RANKS=8 timeout -k 10 200 python tools/pipeline_time.py default4_4k_64spp 96 16 2>/dev/null | tail -n 1 || exit 1
  timeout -k 10 200 python tools/pipeline_time.py glass_1080p_64spp 6 1 2>/dev/null | tail -n 1 || exit 1
  timeout -k 10 200 python tools/pipeline_time.py default4_4k_64spp 20 1 2>/dev/null | tail -n 1 || exit 1
done
