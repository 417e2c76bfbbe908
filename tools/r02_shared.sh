#!/bin/bash
# frames in flight: the direct kernels of all contexts round RT_SEQ_DIRECT_STREAMS shared streams (0: each on its context's
# stream, behind its listed kernel); rank 0's share of an N-GPU frame, 16 contexts, GPU_MAX_HW_QUEUES=16
for k in ${KS:-0 4 5 6 7 8}; do
 echo "== RT_SEQ_DIRECT_STREAMS=$k"
 for n in 8 2 1; do
  for fr in 20 20 20 96; do
   RT_SEQ_DIRECT_STREAMS=$k RT_HW_QUEUES=16 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp $fr 16 2>/dev/null | head -n 1 || exit 1
  done
 done
 RT_SEQ_DIRECT_STREAMS=$k RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py glass_1080p_64spp 32 16 2>/dev/null | head -n 1 || exit 1
done
