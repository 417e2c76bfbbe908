#!/bin/bash
# GPU suite, then the default bench line (no CPU baseline): tools/r03_check.sh [tag]
set -o pipefail
T=${1:-r03}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail gpurun_out/${T}_bench.err; exit 1; }
python3 -c "import json,sys; d=json.load(open('gpurun_out/${T}_bench.json')); print('default', d['ms_per_step'], d['value'], 'latency', d['frame_latency_ms'], 'kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'], d['verified'])"
