#!/bin/bash
# C4 (2049 hittables, 8 point lights): the C4 tests, then occupancy / VALU and HBM traffic per launch.  tools/r03_c4.sh <tag>
set -o pipefail
T=${1:-c4}
timeout -k 10 600 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "c4 or large or cull or coop or solo or many" > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
tools/pmc_quick.sh $T many2048_1080p_32spp 0 && tools/pmc_traffic.sh $T many2048_1080p_32spp 0
