#!/bin/bash
# frames in flight: the hand-over threshold of the sample-parallel route (draws x samples left)
for w in 12000 48000 200000 4000000000; do
  echo "== RT_EXP_HEAVY_MIN_WORK=$w"
  RT_EXP_HEAVY_MIN_WORK=$w timeout -k 10 300 python bench.py --workload glass_1080p_64spp --steps 32 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('glass', d['ms_per_step'])" || exit 1
  RT_EXP_HEAVY_MIN_WORK=$w timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('4k', d['ms_per_step'])" || exit 1
  RT_EXP_HEAVY_MIN_WORK=$w RT_HW_QUEUES=16 RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 96 16 2>/dev/null | tail -n 1 || exit 1
done
