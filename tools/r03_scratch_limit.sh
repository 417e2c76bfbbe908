#!/bin/bash
# frames in flight against the runtime's per-queue scratch limit (experiment): latencies of the render calls and the rate
export RT_HW_QUEUES=16
for lim in default 4294967296; do
  if [ "$lim" = default ]; then unset HSA_SCRATCH_SINGLE_LIMIT HSA_SCRATCH_SINGLE_LIMIT_ASYNC; else export HSA_SCRATCH_SINGLE_LIMIT=$lim HSA_SCRATCH_SINGLE_LIMIT_ASYNC=$lim; fi
  echo "== scratch limit $lim"
  RT_SLOW_CALLS=0 timeout -k 10 120 python tools/pipeline_time.py glass_1080p_64spp 48 16 2>/dev/null > gpurun_out/scratch_$lim.log || exit 1
  grep "ms/frame" gpurun_out/scratch_$lim.log
  grep "slow call" gpurun_out/scratch_$lim.log | awk '{s+=$6; n++; if ($6>m) m=$6} END {printf "  render calls: %d, mean %.1f ms, max %.1f ms\n", n, s/n, m}'
  timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 64 16 2>/dev/null | tail -n 1
done
