#!/bin/bash
# the default bench (20 steps, 16 frames in flight) with the two hand-over bars of sequence mode, alternating
for rep in 1 2 3; do
  for w in 12000 200000; do
    RT_HEAVY_MIN_WORK_SEQ=$w timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w steps20', d['ms_per_step'])" || exit 1
  done
done
for w in 12000 200000; do
  RT_HEAVY_MIN_WORK_SEQ=$w timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w steps100', d['ms_per_step'])" || exit 1
  RT_HEAVY_MIN_WORK_SEQ=$w timeout -k 10 300 python bench.py --workload glass_1080p_64spp --steps 32 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w glass', d['ms_per_step'])" || exit 1
  RT_HEAVY_MIN_WORK_SEQ=$w RT_HW_QUEUES=16 RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 20 16 2>/dev/null | head -n 1 || exit 1
done
