#!/bin/bash
# rays_kernel (flags 32 = RT_FLAG_RAYS_LISTED) against the default route (flags 0): frames alone, rank shares, frames in flight
set -e
for wl in default4_4k_64spp glass_1080p_64spp default4_1080p_16spp; do
  timeout -k 10 120 python tools/time_workload.py $wl 0 32 2>&1 | grep -v "^$"
done
for n in 2 4 8; do
  echo "rank share 1/$n"; RT_EMULATE_RANKS=$n timeout -k 10 100 python tools/time_workload.py default4_4k_64spp 0 32 2>&1 | tail -6
done
echo "in flight"
RT_HW_QUEUES=16 RT_FLAGS=48 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 24 8
RT_HW_QUEUES=16 RT_FLAGS=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 24 8
RT_HW_QUEUES=16 RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 48 8
RT_HW_QUEUES=16 RT_EMULATE_RANKS=8 RT_FLAGS=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 48 8
RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py glass_1080p_64spp 12 6
