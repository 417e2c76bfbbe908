#!/bin/bash
# several C4 frames in one process, per experiment build: tools/c4_reps.sh <suffix> ...
for S in "$@"; do
  echo "== $S"
  RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$S.so timeout -k 10 200 python -u tools/time_workload.py many2048_1080p_32spp 0 0 0 2>&1 | grep wall | awk '{print $5}' | tr "\n" " "; echo
done
