#!/bin/bash
# 20 frames of rank 0's share of an 8-GPU frame (what `bench.py --steps 20` times at N = 8): frames in flight and hardware queues
for q in 16 24; do
  for f in 10 16 20; do
    echo "queues $q"
    RT_HW_QUEUES=$q RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 20 $f 2>/dev/null || exit 1
  done
done
