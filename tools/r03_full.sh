#!/bin/bash
# GPU suite, then the driver's bench command (default flags): tools/r03_full.sh [tag]
set -o pipefail
T=${1:-r03}
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
timeout -k 10 600 python -u bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail gpurun_out/${T}_bench.err; exit 1; }
tail -n 3 gpurun_out/${T}_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench.json'))
print('default', d['ms_per_step'], d['value'], 'latency', d['frame_latency_ms'], 'kernel_ms', d['kernel_ms'], 'roof', {k: d['roofline'].get(k) for k in ('frac','pmc_stale','executed_frac')}, d['verified'])
print('relaxed', d.get('relaxed'))
print('secondary', d.get('secondary'))
print('multi_gpu', d.get('multi_gpu'))
print('cpu', d.get('cpu_baseline'))"
