#!/bin/bash
# do the waiting host threads of the frames in flight cost CPU time (the box's cgroup allows 16 CPUs per 100 ms period)?
TIMEFORMAT="elapsed %R s user %U s sys %S s"
for w in spin sleep; do   # (both set explicitly)
 for n in 8 1; do
  echo "== waits $w, rank 0 of $n"
  time (RT_WAITS=$w RT_HW_QUEUES=16 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp $((n*100)) 16 2>/dev/null) || exit 1
  RT_WAITS=$w RT_HW_QUEUES=16 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 20 16 2>/dev/null || exit 1
 done
 RT_WAITS=$w RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py glass_1080p_64spp 32 16 2>/dev/null || exit 1
 time (RT_BENCH_WAITS=$w timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330) || exit 1
done
