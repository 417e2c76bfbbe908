#!/bin/bash
# A/B of alternative builds of the C-ABI library in one GPU session: tools/ab.sh <workload> <lib suffix>...
WL=$1; shift
for v in "$@"; do
  lib=raytracer_amd/librt_amd_$v.so; [ "$v" = "main" ] && lib=raytracer_amd/librt_amd.so
  for rep in 1 2; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --workload $WL 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['config']['workload'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['kernel_ms'], 'Mrays/s', d['value'])"
  done
done
