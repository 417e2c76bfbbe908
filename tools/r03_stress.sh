#!/bin/bash
# the GPU suite, then the parity sweeps (random frames over every scene and route; larger frames; large scenes)
set -o pipefail
T=${1:-r03}
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
timeout -k 10 200 python -u tests/stress_parity.py ${2:-120} 3 > gpurun_out/${T}_stress.log 2>&1 || { tail -n 20 gpurun_out/${T}_stress.log; exit 1; }
tail -n 2 gpurun_out/${T}_stress.log
RT_STRESS_BIG=1 timeout -k 10 200 python -u tests/stress_parity.py ${3:-90} 4 > gpurun_out/${T}_stress_big.log 2>&1 || { tail -n 20 gpurun_out/${T}_stress_big.log; exit 1; }
tail -n 2 gpurun_out/${T}_stress_big.log
RT_STRESS_LARGE=1 timeout -k 10 200 python -u tests/stress_parity.py ${3:-90} 5 > gpurun_out/${T}_stress_large.log 2>&1 || { tail -n 20 gpurun_out/${T}_stress_large.log; exit 1; }
tail -n 2 gpurun_out/${T}_stress_large.log
