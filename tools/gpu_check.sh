#!/bin/bash
# Runs on the GPU box: the GPU test suite, then bench A/B lines.  usage: tools/gpu_check.sh "<workloads>" "<flags list>"
WLS=${1:-default4_4k_64spp}; FLAGS=${2:-0}
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo exit=$rc >> gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for wl in $WLS; do for f in $FLAGS; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --flags $f --workload $wl 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl flags', $f, 'ms/step', d['ms_per_step'], 'kernel_ms', d['kernel_ms'], 'device_ms', d['device_ms_rank0'], 'Mrays/s', d['value'], 'frac', d['roofline']['frac'])"
done; done
