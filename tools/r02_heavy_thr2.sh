#!/bin/bash
# the default bench (20 steps, 16 frames in flight) with the two hand-over bars of sequence mode, alternating
for rep in 1 2 3; do
  for s in seq12k seq200k; do
    RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$s.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s steps20', d['ms_per_step'])" || exit 1
  done
done
for s in seq12k seq200k; do
  RT_AMD_LIB=$PWD/raytracer_amd/librt_amd_$s.so timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s steps100', d['ms_per_step'])" || exit 1
done
