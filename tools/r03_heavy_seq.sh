#!/bin/bash
# the hand-over bar under RT_FLAG_SEQUENCE (RT_HEAVY_MIN_WORK_SEQ) against frame shares: tools/r03_heavy_seq.sh <bar|default>...
for b in "$@"; do
  if [ "$b" = default ]; then unset RT_HEAVY_MIN_WORK_SEQ; else export RT_HEAVY_MIN_WORK_SEQ=$b; fi
  echo "== RT_HEAVY_MIN_WORK_SEQ=$b"
  for r in 8 4 2 1; do
    RT_EMULATE_RANKS=$r RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp $((r > 1 ? 96 : 64)) 16 2>/dev/null | tail -n 1
  done
  RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 20 16 2>/dev/null | tail -n 1
  RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py glass_1080p_64spp 48 16 2>/dev/null | tail -n 1
done
