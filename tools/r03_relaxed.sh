#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u tools/relaxed_margins.py gpurun_out/r03_relaxed_margins.json || exit 1
for f in 0 128; do
  echo "== flags $f"
  tools/pmc_quick.sh rlx_$f default4_4k_64spp $f | head -2 | cut -c1-60,90-400
  RT_FLAGS=$((f+16)) RT_HW_QUEUES=16 timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 48 16 2>/dev/null | tail -n 1
done
