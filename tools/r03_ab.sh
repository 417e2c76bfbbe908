#!/bin/bash
# A/B of experiment builds (tools/devbuild.sh <suffix>): PMC pass + timing per library.  tools/r03_ab.sh <workload> <suffix>...
WL=$1; shift
for s in "$@"; do
  export RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$s.so
  echo "== $s"
  tools/pmc_quick.sh ab_$s $WL 0 | head -3 | cut -c1-60,90-400
  RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py $WL 48 16 2>/dev/null | tail -n 1
done
