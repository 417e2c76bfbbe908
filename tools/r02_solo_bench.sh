#!/bin/bash
# frames in flight with and without the solo route: glass 1080p and the 4K headline
for w in 0 20000 5000; do
  echo "== RT_SOLO_MIN_WORK=$w"
  RT_SOLO_MIN_WORK=$w timeout -k 10 300 python bench.py --workload glass_1080p_64spp --steps 32 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('glass', d['ms_per_step'], d['frame_latency_ms'])" || exit 1
  RT_SOLO_MIN_WORK=$w timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('4k', d['ms_per_step'], d['frame_latency_ms'])" || exit 1
  RT_SOLO_MIN_WORK=$w RT_HW_QUEUES=16 RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 96 16 2>/dev/null | tail -n 1 || exit 1
done
