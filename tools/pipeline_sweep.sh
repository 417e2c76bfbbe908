#!/bin/bash
# tools/pipeline_sweep.sh <workload> <frames> "<F list>" "<ranks list>" [hw queues]
set -e
wl=${1:-default4_4k_64spp}; frames=${2:-40}; fl=${3:-"1 2 3 4"}; rl=${4:-"1 2 4 8"}; q=${5:-}
mkdir -p gpurun_out
for n in $rl; do
  RT_HW_QUEUES=$q RT_EMULATE_RANKS=$n timeout -k 10 150 python3 tools/pipeline_time.py $wl $frames $fl 2>&1 | tee -a gpurun_out/pipeline_sweep_q${q:-def}.log
done
