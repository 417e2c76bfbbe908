#!/bin/bash
# GPU tests, the default bench, the glass bench in flight, and rank 0's share of an N-GPU frame with 16 frames in flight
set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/all_tests.log 2>&1 || { tail -n 20 gpurun_out/all_tests.log; exit 1; }
tail -n 2 gpurun_out/all_tests.log
timeout -k 10 600 python -u bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail gpurun_out/bench_default.err; exit 1; }
python3 -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print('default', d['ms_per_step'], d['value'], d['frame_latency_ms'], d['kernel_ms'], d['roofline']['frac'], d['roofline'].get('timed_region_valu_issue_frac'), d['verified'], d['cpu_baseline'])"
timeout -k 10 600 python -u bench.py --workload glass_1080p_64spp --steps 32 --no-cpu-baseline > gpurun_out/bench_glass.json 2> gpurun_out/bench_glass.err || { tail gpurun_out/bench_glass.err; exit 1; }
python3 -c "import json; d=json.load(open('gpurun_out/bench_glass.json')); print('glass', d['ms_per_step'], d['value'], d['frame_latency_ms'], d['kernel_ms'], d['verified'])"
for n in 8 4 2 1; do RT_HW_QUEUES=16 RT_EMULATE_RANKS=$n timeout -k 10 120 python tools/pipeline_time.py default4_4k_64spp 96 16 2>/dev/null | tail -n 2 || exit 1; done
