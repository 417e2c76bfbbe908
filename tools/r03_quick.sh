#!/bin/bash
# parity subset + counters + in-flight timing of the default workload: tools/r03_quick.sh <tag>
set -o pipefail
T=${1:-q}
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
tools/pmc_quick.sh $T default4_4k_64spp 0 | head -3 | cut -c1-60,90-400
for i in 1 2; do RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 64 16 2>/dev/null | tail -n 1; done
RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py glass_1080p_64spp 48 16 2>/dev/null | tail -n 1
RT_EMULATE_RANKS=8 RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 96 16 2>/dev/null | tail -n 1
RT_EMULATE_RANKS=8 timeout -k 10 120 python tools/time_workload.py default4_4k_64spp 0 | tail -n 1
timeout -k 10 120 python tools/time_workload.py glass_1080p_64spp 0 | tail -n 1
