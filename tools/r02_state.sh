#!/bin/bash
# where the tree stands: frames alone and in flight, whole frame and rank shares, default route (0) and rays_kernel (32)
for n in 1 2 4 8; do
  echo "== share 1/$n alone (flags 0, 32)"
  RT_EMULATE_RANKS=$n timeout -k 10 100 python tools/time_workload.py default4_4k_64spp 0 32 2>&1 | grep -v amdgpu.ids | awk 'NR%3==0'
  echo "== share 1/$n, 8 in flight (flags 16, 48)"
  RT_HW_QUEUES=16 RT_FLAGS=16 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp $((24*n)) 8 2>&1 | grep -v amdgpu.ids | tail -1
  RT_HW_QUEUES=16 RT_FLAGS=48 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp $((24*n)) 8 2>&1 | grep -v amdgpu.ids | tail -1
done
