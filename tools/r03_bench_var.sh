#!/bin/bash
# run-to-run spread of the driver's bench command (headline only): tools/r03_bench_var.sh <runs>
for i in $(seq 1 ${1:-6}); do
  timeout -k 10 300 python -u bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('run', d['ms_per_step'], d['value'], 'gather max', d['multi_gpu']['gather_ms_rank0']['max'])" || exit 1
done
