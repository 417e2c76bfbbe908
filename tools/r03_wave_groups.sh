#!/bin/bash
# a wave per workgroup for the listed kernel of long frames in flight (RT_WAVE_GROUPS_MS: never / default / always)
export RT_HW_QUEUES=16
for ms in 1e9 default 0 1e9 default; do
  if [ "$ms" = default ]; then unset RT_WAVE_GROUPS_MS; else export RT_WAVE_GROUPS_MS=$ms; fi
  echo "== RT_WAVE_GROUPS_MS=$ms"
  timeout -k 10 200 python tools/pipeline_time.py glass_1080p_64spp 64 16 2>/dev/null | tail -n 1 || exit 1
  timeout -k 10 200 python tools/pipeline_time.py default4_4k_64spp 64 16 2>/dev/null | tail -n 1 || exit 1
  RT_EMULATE_RANKS=8 timeout -k 10 200 python tools/pipeline_time.py default4_4k_64spp 96 16 2>/dev/null | tail -n 1 || exit 1
  timeout -k 10 200 python tools/pipeline_time.py default4_4k_64spp 20 1 2>/dev/null | tail -n 1 || exit 1
done
