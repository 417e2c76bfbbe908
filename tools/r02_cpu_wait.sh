#!/bin/bash
# what the waiting host threads of 16 frames in flight cost: RT_WAIT_SLEEP_US = 0 (hipStreamSynchronize, polling without
# pause), 40 (the default: an event polled with 40 us naps), 200
TIMEFORMAT="elapsed %R s user %U s sys %S s"
for us in 0 40 200; do
 for n in 8 1; do
  echo "== RT_WAIT_SLEEP_US=$us, rank 0 of $n"
  time (RT_WAIT_SLEEP_US=$us RT_HW_QUEUES=16 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp $((n*100)) 16 2>/dev/null) || exit 1
  RT_WAIT_SLEEP_US=$us RT_HW_QUEUES=16 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 20 16 2>/dev/null || exit 1
 done
 RT_WAIT_SLEEP_US=$us RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py glass_1080p_64spp 32 16 2>/dev/null || exit 1
done
