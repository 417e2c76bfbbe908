#!/bin/bash
# with the direct kernels on shared streams: more contexts in flight for the eighth share?
for f in 16 20 24 32; do
  for fr in 20 96; do
    RT_HW_QUEUES=16 RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp $fr $f 2>/dev/null | head -n 1 || exit 1
  done
done
for f in 20 24; do
  RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 96 $f 2>/dev/null | head -n 1 || exit 1
done
