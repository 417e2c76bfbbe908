#!/bin/bash
# the solo route (long unsteady pixels -> a wave each): parity sweep with the test hook, then timings alone
timeout -k 10 200 python -u tests/stress_parity.py 60 43 > gpurun_out/solo_stress.log 2>&1 || { tail -n 5 gpurun_out/solo_stress.log; exit 1; }
tail -n 1 gpurun_out/solo_stress.log
RT_STRESS_LARGE=1 timeout -k 10 200 python -u tests/stress_parity.py 40 44 > gpurun_out/solo_stress_large.log 2>&1 || { tail -n 5 gpurun_out/solo_stress_large.log; exit 1; }
tail -n 1 gpurun_out/solo_stress_large.log
for cfg in "0 64 512" "20000 64 512" "5000 64 512" "2000 64 512" "5000 1280 8192" "2000 1280 8192"; do
  set -- $cfg
  echo "== RT_SOLO_MIN_WORK=$1 pool per $2 max $3 MB"
  RT_SOLO_MIN_WORK=$1 RT_EXP_POOL_PER=$2 RT_EXP_POOL_MAX_MB=$3 timeout -k 10 120 python tools/time_workload.py glass_1080p_64spp 0 2>/dev/null | tail -n 1 || exit 1
  RT_SOLO_MIN_WORK=$1 RT_EXP_POOL_PER=$2 RT_EXP_POOL_MAX_MB=$3 RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/time_workload.py default4_4k_64spp 0 2>/dev/null | tail -n 1 || exit 1
done
RT_SOLO_MIN_WORK=5000 RT_EXP_POOL_PER=1280 RT_EXP_POOL_MAX_MB=8192 bash tools/r02_timeline.sh solo glass_1080p_64spp 0 2>&1 | tail -n 9
